"""yacs-free experiment configuration with the reference's keys and defaults.

Mirrors the key tree of the reference ``lib/config/defaults.py:1-144`` (a yacs
``CfgNode``) so the shipped ``configs/cuhkpedes/*.yaml`` files merge unchanged;
``merge_from_file`` / ``merge_from_list`` / ``freeze`` behave like yacs for the
subset the launchers use (train_net.py:156-159).  yacs is not available in the
build image; PyYAML is.
"""

import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError("config is frozen: cannot set %s" % k)
        self[k] = v

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def defrost(self):
        self.__dict__["_frozen"] = False
        for v in self.values():
            if isinstance(v, CfgNode):
                v.defrost()

    def clone(self):
        return CfgNode(copy.deepcopy(dict(self)))

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError("non-existent config key: %s%s" % (path, k))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise TypeError("%s%s must be a mapping" % (path, k))
                self[k]._merge(v, path + k + ".")
            else:
                self[k] = _coerce(v, self[k], path + k)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        opts = list(opts or [])
        assert len(opts) % 2 == 0, "opts must be KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("non-existent config key: %s" % key)
            if isinstance(val, str):
                try:
                    val = ast.literal_eval(val)
                except (ValueError, SyntaxError):
                    pass
            node[parts[-1]] = _coerce(val, node[parts[-1]], key)


def _coerce(new, old, key):
    if isinstance(old, tuple) and isinstance(new, (list, str)):
        new = tuple(ast.literal_eval(new)) if isinstance(new, str) else tuple(new)
    if isinstance(old, list) and isinstance(new, tuple):
        new = list(new)
    if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
        new = float(new)
    if old is not None and new is not None and type(old) is not type(new):
        raise TypeError("type mismatch for %s: %r vs default %r" % (key, new, old))
    return new


_DEFAULTS = {
    "ROOT": "./",
    "DATASETS": {"TRAIN": (), "TEST": (), "USE_ONEHOT": True},
    "DATALOADER": {"NUM_WORKERS": 4, "IMS_PER_ID": 4, "EN_SAMPLER": True},
    "INPUT": {"HEIGHT": 224, "WIDTH": 224, "PIXEL_MEAN": [0.485, 0.456, 0.406], "PIXEL_STD": [0.229, 0.224, 0.225],
              "PADDING": 10, "USE_AUG": False},
    "MODEL": {
        "DEVICE": "cuda", "VISUAL_MODEL": "resnet50", "TEXTUAL_MODEL": "bilstm", "NUM_CLASSES": 11003,
        "FREEZE": False, "WEIGHT": "imagenet",
        "MOCO": {"K": 1024, "M": 0.999, "FC": True},
        "GRU": {"ONEHOT": "yes", "EMBEDDING_SIZE": 512, "NUM_UNITS": 512, "VOCABULARY_SIZE": 12000,
                "DROPOUT_KEEP_PROB": 0.7, "MAX_LENGTH": 100, "NUM_LAYER": 1},
        "RESNET": {"RES5_STRIDE": 2, "RES5_DILATION": 1, "PRETRAINED": None},
        "EMBEDDING": {"EMBED_HEAD": "simple", "FEATURE_SIZE": 512, "DROPOUT_PROB": 0.3, "EPSILON": 0.0},
    },
    "SOLVER": {
        "IMS_PER_BATCH": 16, "NUM_EPOCHS": 100, "CHECKPOINT_PERIOD": 1, "EVALUATE_PERIOD": 1, "OPTIMIZER": "Adam",
        "BASE_LR": 0.0002, "BIAS_LR_FACTOR": 2, "WEIGHT_DECAY": 0.00004, "WEIGHT_DECAY_BIAS": 0.0, "ADAM_ALPHA": 0.9,
        "ADAM_BETA": 0.999, "SGD_MOMENTUM": 0.9, "LRSCHEDULER": "step", "WARMUP_FACTOR": 1.0 / 3, "WARMUP_EPOCHS": 10,
        "WARMUP_METHOD": "linear", "GAMMA": 0.1, "STEPS": (500,), "POWER": 0.9, "TARGET_LR": 0.0001,
    },
    "TEST": {"IMS_PER_BATCH": 16},
    "DTYPE": "float32",
    "AMP_VERBOSE": False,
}


def get_cfg_defaults():
    return CfgNode(copy.deepcopy(_DEFAULTS))


def moco_cfg(visual="m_resnet50", K=2048, height=384, width=128, num_classes=11003):
    """The shipped ``moco_gru_clip{rn50,rn101}_ls_bs128_2048.yaml`` settings as a
    CfgNode (for benchmarks and tests where the YAML file is not on disk)."""
    cfg = get_cfg_defaults()
    cfg.MODEL.VISUAL_MODEL = visual
    cfg.MODEL.TEXTUAL_MODEL = "bigru"
    cfg.MODEL.NUM_CLASSES = num_classes
    cfg.MODEL.GRU.ONEHOT = "clip_vit"
    cfg.MODEL.GRU.VOCABULARY_SIZE = 512
    cfg.MODEL.GRU.DROPOUT_KEEP_PROB = 1.0
    cfg.MODEL.RESNET.RES5_STRIDE = 1
    cfg.MODEL.EMBEDDING.EMBED_HEAD = "moco"
    cfg.MODEL.EMBEDDING.FEATURE_SIZE = 256
    cfg.MODEL.EMBEDDING.DROPOUT_PROB = 0.0
    cfg.MODEL.EMBEDDING.EPSILON = 0.1
    cfg.MODEL.MOCO.FC = False
    cfg.MODEL.MOCO.K = K
    cfg.INPUT.HEIGHT, cfg.INPUT.WIDTH = height, width
    cfg.SOLVER.IMS_PER_BATCH = 128
    cfg.SOLVER.BASE_LR = 0.0001
    return cfg
