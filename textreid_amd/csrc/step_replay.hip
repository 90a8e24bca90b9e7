// The recorded train step, re-issued as ordinary stream launches from ONE host call (counterpart of the loop body
// lib/engine/trainer.py:72-91 once engine/graph.py has recorded it).
//
// A step is ~1100 kernel launches on four streams; issued from Python through ctypes they cost 35-39 ms of host time per
// 43 ms step, and the order in which the host reaches the side streams decides how much of the key encoder / text encoders
// overlaps the query encoder.  hipGraphLaunch of the recorded step removes the host work but runs SLOWER than the eager
// step on this runtime (45.2 vs 44.0 ms: profiles/r04l_*).  This file takes the third road: it reads the recorded hipGraph
// back (nodes, launch parameters, edges), lays the nodes out on a handful of streams so that chains stay in stream order and
// only cross-chain edges need an event, and replays them with hipLaunchKernel - the same launches the eager step makes, with
// every stream fed within a few milliseconds.  The graph object stays alive (torch.cuda.CUDAGraph(keep_graph=True)): the
// argument buffers the launches point at are the graph's own copies.
//
// Data parallel (lib/engine/trainer.py:72-91 under train_net.py:50-56): the step's collectives - the packed embedding
// all-gather, the gradient all-reduces staged from inside backward and the buckets after it - are NOT recorded.  The capture
// puts a MARKER (the empty kernel below, its argument the collective's index) where each one belongs; the plan treats a marker
// as a cut point: trid_step_replay_run_segment re-issues the nodes up to the next marker, hands the marker's stream back to the
// host, which enqueues the collective there through the communication library's own launch path (RCCL's kernels are never
// re-issued from read-back launch parameters), and continues.  At most ~8 host round trips per step.

#include <stdlib.h>

#include <algorithm>
#include <queue>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace trid {

// where a collective belongs in the recording (never launched by the replay: its slot is the host's)
__global__ void step_marker_kernel(int id) { (void)id; }

namespace {

struct RNode {
    hipGraphNodeType type;
    hipKernelNodeParams k;
    hipMemcpy3DParms cp;
    hipMemsetParams ms;
    int lane = -1;
    int record = -1;         // event recorded behind this node (it has a successor on another lane), or -1
    int marker = -1;         // >= 0: a cut point (index of the collective the host issues here)
    std::vector<int> waits;  // events of predecessors on other lanes
};

struct Replay {
    std::vector<RNode> nodes;   // in issue order
    std::vector<hipStream_t> lanes;
    std::vector<hipEvent_t> events;
    hipEvent_t fork = nullptr;  // recorded on the caller's stream; every lane waits for it first
    std::vector<hipEvent_t> joins;
    int n_kernels = 0, n_copies = 0, n_sets = 0, n_empty = 0, n_markers = 0;
    int device = -1;            // the device the lanes and events belong to
    size_t cursor = 0;          // next node to issue (a step in progress: segmented runs)
    bool in_step = false, at_marker = false;
    bool poisoned = false;      // a launch failed in mid-step: the handle refuses further runs
};

void destroy(Replay* r) {
    if (r == nullptr) return;
    for (hipEvent_t e : r->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->joins) (void)hipEventDestroy(e);
    if (r->fork) (void)hipEventDestroy(r->fork);
    for (hipStream_t s : r->lanes) (void)hipStreamDestroy(s);
    delete r;
}

#define RP_CHECK(call, what)                                                        \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            set_error("trid_step_replay: %s: %s", what, hipGetErrorString(e_));     \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

}  // namespace
}  // namespace trid

using namespace trid;

extern "C" int trid_step_replay_build(void* graph_, int max_lanes, void** out) {
    TRID_REQUIRE(graph_ && out && max_lanes >= 1 && max_lanes <= 32, "trid_step_replay_build: bad arguments");
    hipGraph_t graph = (hipGraph_t)graph_;
    *out = nullptr;
    size_t n = 0, ne = 0;
    RP_CHECK(hipGraphGetNodes(graph, nullptr, &n), "hipGraphGetNodes");
    TRID_REQUIRE(n > 0, "trid_step_replay_build: empty graph");
    std::vector<hipGraphNode_t> gn(n);
    RP_CHECK(hipGraphGetNodes(graph, gn.data(), &n), "hipGraphGetNodes");
    RP_CHECK(hipGraphGetEdges(graph, nullptr, nullptr, &ne), "hipGraphGetEdges");
    std::vector<hipGraphNode_t> ef(ne), et(ne);
    if (ne) RP_CHECK(hipGraphGetEdges(graph, ef.data(), et.data(), &ne), "hipGraphGetEdges");
    std::unordered_map<hipGraphNode_t, int> index;
    for (size_t i = 0; i < n; ++i) index[gn[i]] = (int)i;
    std::vector<std::vector<int>> preds(n), succs(n);
    for (size_t e = 0; e < ne; ++e) {
        auto a = index.find(ef[e]), b = index.find(et[e]);
        TRID_REQUIRE(a != index.end() && b != index.end(), "trid_step_replay_build: edge to an unknown node");
        preds[b->second].push_back(a->second);
        succs[a->second].push_back(b->second);
    }
    // issue order: topological, ties by the node's position in the recording (the order the eager step issued them)
    std::vector<int> indeg(n), order;
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (size_t i = 0; i < n; ++i) {
        indeg[i] = (int)preds[i].size();
        if (indeg[i] == 0) ready.push((int)i);
    }
    while (!ready.empty()) {
        const int v = ready.top();
        ready.pop();
        order.push_back(v);
        for (int s : succs[v])
            if (--indeg[s] == 0) ready.push(s);
    }
    TRID_REQUIRE(order.size() == n, "trid_step_replay_build: the graph has a cycle");

    Replay* r = new Replay();
    std::vector<RNode> raw(n);
    for (RNode& nd : raw) { memset(&nd.k, 0, sizeof(nd.k)); memset(&nd.cp, 0, sizeof(nd.cp)); memset(&nd.ms, 0, sizeof(nd.ms)); }
    for (size_t i = 0; i < n; ++i) {
        RNode& nd = raw[i];
        hipError_t e = hipGraphNodeGetType(gn[i], &nd.type);
        if (e == hipSuccess) {
            switch (nd.type) {
                case hipGraphNodeTypeKernel:
                    e = hipGraphKernelNodeGetParams(gn[i], &nd.k);
                    r->n_kernels++;
                    if (e == hipSuccess && nd.k.func == (void*)step_marker_kernel) {
                        if (nd.k.kernelParams == nullptr || nd.k.kernelParams[0] == nullptr) {
                            set_error("trid_step_replay_build: a marker node without readable arguments");
                            destroy(r);
                            return TRID_E_UNSUPPORTED;
                        }
                        nd.marker = *reinterpret_cast<const int*>(nd.k.kernelParams[0]);
                        r->n_markers++;
                    }
                    break;
                case hipGraphNodeTypeMemcpy: e = hipGraphMemcpyNodeGetParams(gn[i], &nd.cp); r->n_copies++; break;
                case hipGraphNodeTypeMemset: e = hipGraphMemsetNodeGetParams(gn[i], &nd.ms); r->n_sets++; break;
                case hipGraphNodeTypeEmpty: r->n_empty++; break;
                default:
                    set_error("trid_step_replay_build: node type %d cannot be replayed as a stream launch", (int)nd.type);
                    destroy(r);
                    return TRID_E_UNSUPPORTED;
            }
        }
        if (e != hipSuccess) {
            set_error("trid_step_replay_build: reading node %d failed: %s", (int)i, hipGetErrorString(e));
            destroy(r);
            return (int)e;
        }
        if (nd.type == hipGraphNodeTypeMemcpy) {
            // (this runtime returns nothing for the 1-D copy nodes a captured hipMemcpyAsync leaves: refuse rather than guess)
            const hipMemcpy3DParms& c = nd.cp;
            if (c.srcPtr.ptr == nullptr || c.dstPtr.ptr == nullptr || c.extent.width == 0) {
                if (getenv("TRID_REPLAY_DEBUG")) {
                    for (int pass = 0; pass < 2; ++pass)
                        for (int q : (pass ? succs[i] : preds[i])) {
                            hipGraphNodeType t;
                            hipKernelNodeParams kp;
                            if (hipGraphNodeGetType(gn[q], &t) == hipSuccess && t == hipGraphNodeTypeKernel &&
                                hipGraphKernelNodeGetParams(gn[q], &kp) == hipSuccess)
                                fprintf(stderr, "[step_replay] copy node %d: %s kernel %s\n", (int)i, pass ? "followed by" : "follows", hipKernelNameRefByPtr(kp.func, nullptr));
                        }
                }
                set_error("trid_step_replay_build: copy node %d cannot be read back (a captured hipMemcpyAsync)", (int)i);
                destroy(r);
                return TRID_E_UNSUPPORTED;
            }
        }
        if (nd.type == hipGraphNodeTypeMemset && !(nd.ms.height <= 1 && (nd.ms.elementSize == 1 || nd.ms.elementSize == 2 || nd.ms.elementSize == 4))) {
            set_error("trid_step_replay_build: 2-D memset nodes are not supported");
            destroy(r);
            return TRID_E_UNSUPPORTED;
        }
    }
    // lanes: a node continues the lane of a predecessor that is still that lane's last node (a chain stays in stream
    // order, no event); otherwise it opens a lane while there are lanes left, else joins the lane of its first predecessor
    // (roots: the least loaded lane).  Edges between lanes become one event per producing node.
    std::vector<int> lane(n, -1), tail;  // tail[l] = last node placed on lane l
    std::vector<int> load;
    for (int v : order) {
        int pick = -1;
        for (int p : preds[v])
            if (tail[lane[p]] == p) { pick = lane[p]; break; }
        if (pick < 0) {
            if ((int)tail.size() < max_lanes) {
                pick = (int)tail.size();
                tail.push_back(-1);
                load.push_back(0);
            } else if (!preds[v].empty()) {
                pick = lane[preds[v][0]];
            } else {
                pick = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            }
        }
        lane[v] = pick;
        tail[pick] = v;
        load[pick]++;
    }
    if (getenv("TRID_REPLAY_DEBUG")) {  // what each lane starts with
        for (size_t l = 0; l < tail.size(); ++l) {
            int shown = 0;
            fprintf(stderr, "[step_replay] lane %d: %d nodes:", (int)l, load[l]);
            for (int v : order)
                if (lane[v] == (int)l && raw[v].type == hipGraphNodeTypeKernel && shown < 6) {
                    const char* nm = hipKernelNameRefByPtr(raw[v].k.func, nullptr);
                    fprintf(stderr, " %.28s", nm ? nm : "?");
                    ++shown;
                }
            fprintf(stderr, "\n");
        }
    }
    // (Holding the key encoder a fixed number of kernels behind the query encoder - extra edges b_j after a_{j + d1}, a_{j + d2}
    // after b_j, so that one encoder's BatchNorm passes meet the other's convolutions instead of both running the same kind of
    // kernel at the same time - was tried on this plan: 44.3-45.7 ms per step against 41.85, the ~380 extra event waits cost
    // more than the pairing buys: profiles/r05g_replay_pingpong.txt, tools/exp/r05_run27.sh.  A single extra edge that starts the
    // key encoder k kernels after the query encoder: 41.2 (k = 8) ... 42.1 ms (k = 120) against 41.0: r05g_replay_offset.txt)
    std::vector<int> ev_of(n, -1);
    int n_events = 0;
    for (int v : order)
        for (int p : preds[v])
            if (lane[p] != lane[v] && ev_of[p] < 0) ev_of[p] = n_events++;
    r->nodes.reserve(n);
    // a wait is redundant when an earlier node of the same lane already waited for the same (or a later) node of that lane
    std::vector<std::vector<int>> seen(tail.size(), std::vector<int>(tail.size(), -1));  // seen[lane][other lane] = latest position waited for
    std::vector<int> pos(n, 0);
    for (size_t i = 0; i < order.size(); ++i) pos[order[i]] = (int)i;
    for (int v : order) {
        RNode nd = raw[v];
        nd.lane = lane[v];
        nd.record = ev_of[v];
        for (int p : preds[v]) {
            if (lane[p] == lane[v]) continue;
            if (seen[lane[v]][lane[p]] >= pos[p]) continue;
            seen[lane[v]][lane[p]] = pos[p];
            nd.waits.push_back(ev_of[p]);
        }
        r->nodes.push_back(std::move(nd));
    }
    hipError_t e = hipSuccess;
    // (stream priorities were tried - the recording stream's chain, the step's critical path, on the most urgent priority and
    // the side lanes on the least: 48.8 ms per step against 43.6, the reverse 52.6: tools/exp/r05_run15.sh - and dropped)
    for (size_t l = 0; l < tail.size() && e == hipSuccess; ++l) {
        hipStream_t s = nullptr;
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e == hipSuccess) r->lanes.push_back(s);
    }
    for (int i = 0; i < n_events && e == hipSuccess; ++i) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) r->events.push_back(ev);
    }
    for (size_t l = 0; l < tail.size() && e == hipSuccess; ++l) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) r->joins.push_back(ev);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipGetDevice(&r->device);
    if (e != hipSuccess) {
        set_error("trid_step_replay_build: creating streams / events failed: %s", hipGetErrorString(e));
        destroy(r);
        return (int)e;
    }
    *out = r;
    return TRID_OK;
}

extern "C" int trid_step_marker(int id, void* stream) {
    TRID_REQUIRE(id >= 0, "trid_step_marker: negative id");
    hipLaunchKernelGGL(step_marker_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, id);
    return check_launch("trid_step_marker");
}

extern "C" int trid_step_replay_markers(void* handle) {
    TRID_REQUIRE(handle, "trid_step_replay_markers: null handle");
    return ((const Replay*)handle)->n_markers;
}

extern "C" int trid_step_replay_info(void* handle, int* counts) {
    TRID_REQUIRE(handle && counts, "trid_step_replay_info: null pointer");
    const Replay* r = (const Replay*)handle;
    counts[0] = (int)r->nodes.size();
    counts[1] = r->n_kernels;
    counts[2] = r->n_copies;
    counts[3] = r->n_sets;
    counts[4] = (int)r->lanes.size();
    counts[5] = (int)r->events.size();
    int waits = 0;
    for (const RNode& nd : r->nodes) waits += (int)nd.waits.size();
    counts[6] = waits;
    counts[7] = r->n_empty;
    return TRID_OK;
}

namespace trid {
namespace {

int issue_node(Replay* r, RNode& nd) {
    hipStream_t s = r->lanes[nd.lane];
    switch (nd.type) {
        case hipGraphNodeTypeKernel: {
            hipError_t e;
            if (nd.k.kernelParams != nullptr) {
                e = hipLaunchKernel(nd.k.func, nd.k.gridDim, nd.k.blockDim, nd.k.kernelParams, nd.k.sharedMemBytes, s);
            } else {
                e = hipModuleLaunchKernel((hipFunction_t)nd.k.func, nd.k.gridDim.x, nd.k.gridDim.y, nd.k.gridDim.z, nd.k.blockDim.x,
                                          nd.k.blockDim.y, nd.k.blockDim.z, nd.k.sharedMemBytes, s, nullptr, nd.k.extra);
            }
            RP_CHECK(e, "kernel launch");
            break;
        }
        case hipGraphNodeTypeMemcpy: {
            // (the recording's copies are linear device-to-device copies: a 1-D copy reads back as a one-row, one-slice extent)
            const hipMemcpy3DParms& c = nd.cp;
            const bool linear = c.srcArray == nullptr && c.dstArray == nullptr && c.extent.height <= 1 && c.extent.depth <= 1 &&
                                c.srcPos.x == 0 && c.srcPos.y == 0 && c.srcPos.z == 0 && c.dstPos.x == 0 && c.dstPos.y == 0 && c.dstPos.z == 0;
            if (linear) RP_CHECK(hipMemcpyAsync(c.dstPtr.ptr, c.srcPtr.ptr, c.extent.width, c.kind, s), "hipMemcpyAsync");
            else {
                hipError_t e = hipMemcpy3DAsync(&nd.cp, s);
                if (e != hipSuccess) {
                    set_error("trid_step_replay: hipMemcpy3DAsync: %s (extent %zu x %zu x %zu, src %p pitch %zu pos %zu,%zu,%zu, dst %p pitch %zu pos %zu,%zu,%zu, "
                              "arrays %p %p, kind %d)", hipGetErrorString(e), c.extent.width, c.extent.height, c.extent.depth, c.srcPtr.ptr, c.srcPtr.pitch,
                              c.srcPos.x, c.srcPos.y, c.srcPos.z, c.dstPtr.ptr, c.dstPtr.pitch, c.dstPos.x, c.dstPos.y, c.dstPos.z, (void*)c.srcArray,
                              (void*)c.dstArray, (int)c.kind);
                    return (int)e;
                }
            }
            break;
        }
        case hipGraphNodeTypeMemset:
            if (nd.ms.elementSize == 1) RP_CHECK(hipMemsetAsync(nd.ms.dst, (int)nd.ms.value, nd.ms.width, s), "hipMemsetAsync");
            else if (nd.ms.elementSize == 2) RP_CHECK(hipMemsetD16Async((hipDeviceptr_t)nd.ms.dst, (unsigned short)nd.ms.value, nd.ms.width, s), "hipMemsetD16Async");
            else RP_CHECK(hipMemsetD32Async((hipDeviceptr_t)nd.ms.dst, (int)nd.ms.value, nd.ms.width, s), "hipMemsetD32Async");
            break;
        default:
            break;  // empty node: only its edges matter
    }
    return TRID_OK;
}

// every lane joined back into `origin`: whatever was issued stays ordered in front of the caller's later work
int join_lanes(Replay* r, hipStream_t origin) {
    int rc = TRID_OK;
    for (size_t l = 0; l < r->lanes.size(); ++l) {
        hipError_t e = hipEventRecord(r->joins[l], r->lanes[l]);
        if (e == hipSuccess) e = hipStreamWaitEvent(origin, r->joins[l], 0);
        if (e != hipSuccess && rc == TRID_OK) {
            set_error("trid_step_replay: joining lane %d: %s", (int)l, hipGetErrorString(e));
            rc = (int)e;
        }
    }
    return rc;
}

// A failed launch / wait / record in mid-step: part of the step is already enqueued on the lanes.  Keep the caller's stream ordered
// behind it (join), and refuse every further run - the optimizer tables were advanced for a step that did not happen as recorded
// (engine/graph.py drops the handle and goes on eagerly).  The error message of the failure is kept.
int fail_step(Replay* r, hipStream_t origin, int rc) {
    char msg[512];
    snprintf(msg, sizeof(msg), "%s", trid_last_error_string());
    (void)join_lanes(r, origin);
    r->poisoned = true;
    r->in_step = r->at_marker = false;
    r->cursor = 0;
    set_error("%s (the step was abandoned in mid-replay; the handle is poisoned)", msg);
    return rc;
}

// the step from the cursor to the next cut point (stop_at_markers) or to its end.  Returns 1 at a marker (*marker_id, *lane set:
// the host enqueues the collective on that stream and calls again), 0 when the step is complete.
int run_some(Replay* r, hipStream_t origin, bool stop_at_markers, int* marker_id, void** lane) {
    if (r->poisoned) {
        set_error("trid_step_replay: this handle was poisoned by a failed replay");
        return TRID_E_INVALID;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != r->device) {
        set_error("trid_step_replay: the current device (%d) is not the one the plan was built on (%d)", dev, r->device);
        return TRID_E_INVALID;
    }
    if (!r->in_step) {
        RP_CHECK(hipEventRecord(r->fork, origin), "hipEventRecord");
        for (hipStream_t s : r->lanes) {
            hipError_t e = hipStreamWaitEvent(s, r->fork, 0);
            if (e != hipSuccess) {
                set_error("trid_step_replay: hipStreamWaitEvent: %s", hipGetErrorString(e));
                return fail_step(r, origin, (int)e);
            }
        }
        r->in_step = true;
        r->cursor = 0;
        r->at_marker = false;
    }
    while (r->cursor < r->nodes.size()) {
        RNode& nd = r->nodes[r->cursor];
        hipStream_t s = r->lanes[nd.lane];
        if (!r->at_marker) {
            for (int w : nd.waits) {
                hipError_t e = hipStreamWaitEvent(s, r->events[w], 0);
                if (e != hipSuccess) {
                    set_error("trid_step_replay: hipStreamWaitEvent: %s", hipGetErrorString(e));
                    return fail_step(r, origin, (int)e);
                }
            }
            if (nd.marker >= 0 && stop_at_markers) {
                r->at_marker = true;  // (the host's collective takes the node's place; the next call records the node's event)
                *marker_id = nd.marker;
                *lane = (void*)s;
                return 1;
            }
            if (nd.marker < 0) {
                const int rc = issue_node(r, nd);
                if (rc != TRID_OK) return fail_step(r, origin, rc);
            }
        }
        r->at_marker = false;
        if (nd.record >= 0) {
            hipError_t e = hipEventRecord(r->events[nd.record], s);
            if (e != hipSuccess) {
                set_error("trid_step_replay: hipEventRecord: %s", hipGetErrorString(e));
                return fail_step(r, origin, (int)e);
            }
        }
        r->cursor++;
    }
    r->in_step = false;
    r->cursor = 0;
    const int rc = join_lanes(r, origin);
    if (rc != TRID_OK) r->poisoned = true;
    return rc;
}

}  // namespace
}  // namespace trid

// Everything already enqueued on `origin` happens before the step; everything enqueued on it afterwards happens after.
extern "C" int trid_step_replay_run(void* handle, void* origin_) {
    TRID_REQUIRE(handle, "trid_step_replay_run: null handle");
    Replay* r = (Replay*)handle;
    TRID_REQUIRE(r->n_markers == 0, "trid_step_replay_run: the recording has %d cut points (collectives): use trid_step_replay_run_segment", r->n_markers);
    TRID_REQUIRE(!r->in_step, "trid_step_replay_run: a segmented step is in progress");
    int id = -1;
    void* lane = nullptr;
    return run_some(r, (hipStream_t)origin_, false, &id, &lane);
}

extern "C" int trid_step_replay_run_segment(void* handle, void* origin_, int* marker_id, void** lane_stream) {
    TRID_REQUIRE(handle && marker_id && lane_stream, "trid_step_replay_run_segment: null pointer");
    *marker_id = -1;
    *lane_stream = nullptr;
    return run_some((Replay*)handle, (hipStream_t)origin_, true, marker_id, lane_stream);
}

extern "C" int trid_step_replay_destroy(void* handle) {
    destroy((Replay*)handle);
    return TRID_OK;
}
