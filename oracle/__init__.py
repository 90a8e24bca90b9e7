"""CPU oracle for the TextReID encode-and-match hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``textreid_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker / reported baseline,
never as the thing measured or shipped.

It is a functional (state-dict in, tensors out) fp32 restatement of the
reference algorithm on plain PyTorch-CPU ops.  The arithmetic of the reference
lives in a third-party dependency that is not vendored under /root/reference:
PyTorch, pinned ``torch==1.10.0`` / ``torchvision==0.11.1``
(reference ``requirements.txt:17-18``).  The reference ships no tests and no
golden vectors for this path, so parity is pinned by fixtures captured from
importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); see DESIGN.md.

Every function cites the reference file:line it follows.
"""
