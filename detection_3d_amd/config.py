"""Config surface of tools/train_net_sparse3d.py for the hot path: the keys of
maskrcnn_benchmark/config/defaults.py that the 3-D detector reads, the three target configs
(configs/{4c,6c,3G6c}), yaml / opts merging, and the derived fields that `intact_cfg`
(tools/train_net_sparse3d.py:231-323) writes before freezing.  yacs is not installed, so CfgNode is
a small attribute dictionary with the same merge semantics."""
import ast
import copy

import numpy as np
import yaml

CLASS_ORDER = ['background', 'wall', 'window', 'door', 'floor', 'ceiling', 'room']  # suncg_metas.py:3-11


def _listify(v):
    if isinstance(v, (tuple, list)):
        return [_listify(x) for x in v]
    return v


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                if k not in self:
                    raise KeyError(f"unknown config section {k}")
                self[k].merge_from_dict(v)
            else:
                if k not in self:
                    raise KeyError(f"unknown config key {k}")
                if isinstance(v, str) and not isinstance(self[k], str):
                    v = ast.literal_eval(v)        # yacs: "(0, -1.57)" in yaml is a python literal
                self[k] = _listify(v)

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError(f"unknown config key {key}")
            node[parts[-1]] = yaml.safe_load(val) if isinstance(val, str) else val

    def clone(self):
        return copy.deepcopy(self)


_DEFAULTS = {
    'DEBUG': {'eval_in_train': 10, 'eval_in_train_per_iter': -1},
    'INPUT': {'CLASSES': ['background', 'wall'], 'SCENES': [], 'ELEMENTS': ['xyz', 'color', 'normal']},
    'MODEL': {
        'DEVICE': 'cuda', 'META_ARCHITECTURE': 'SparseRCNN', 'WEIGHT': '', 'RPN_ONLY': False, 'MASK_ON': False,
        'SEPARATE_CLASSES': [], 'SEPARATE_RPN': True, 'SEPARATE_CLASSES_ID': [],
        'BACKBONE': {'CONV_BODY': 'Sparse-R-50-FPN', 'OUT_CHANNELS': 128},
        'RPN': {
            'ANCHOR_SIZES_3D': [[0.1, 0.3, 3], [0.2, 0.6, 3], [0.4, 1.2, 3], [0.8, 2.4, 3]],
            'YAWS': [0, -1.57, -0.785, 0.785], 'RATIOS': [[1, 1, 1], [1, 2, 1], [2, 1, 1], [1.5, 1.5, 1]],
            'USE_YAWS': [1, 1, 1, 1], 'USE_FPN': True, 'STRADDLE_THRESH': 0,
            'FG_IOU_THRESHOLD': 0.55, 'BG_IOU_THRESHOLD': 0.25, 'YAW_THRESHOLD': 0.7,
            'BATCH_SIZE_PER_IMAGE': 256, 'POSITIVE_FRACTION': 0.5,
            'NMS_THRESH': 0.5, 'NMS_AUG_THICKNESS_Y_Z': [0.3, 0.3],
            'LABEL_AUG_THICKNESS_Y_TAR_ANC': [0.4, 0], 'LABEL_AUG_THICKNESS_Z_TAR_ANC': [0.8, 0],
            'MIN_SIZE': 0, 'FPN_PRE_NMS_TOP_N_TRAIN': 2000, 'FPN_PRE_NMS_TOP_N_TEST': 2000,
            'FPN_POST_NMS_TOP_N_TRAIN': 1000, 'FPN_POST_NMS_TOP_N_TEST': 1000,
            'RPN_HEAD': 'SingleConvRPNHead_Sparse3D', 'RPN_SCALES_FROM_TOP': [4, 3, 2],
            'RPN_3D_2D_SELECTOR': [1, 3, 4, 5], 'ADD_GT_PROPOSALS': True,
            'ANCHOR_STRIDE': [], 'RPN_MAP_SIZES': [],
        },
        'ROI_HEADS': {
            'USE_FPN': True, 'FG_IOU_THRESHOLD': 0.5, 'BG_IOU_THRESHOLD': 0.5,
            'BBOX_REG_WEIGHTS': [1., 1, 1, 1, 1, 1, 1], 'BATCH_SIZE_PER_IMAGE': 512, 'POSITIVE_FRACTION': 0.25,
            'SCORE_THRESH': 0.05, 'NMS': 0.45, 'NMS_AUG_THICKNESS_Y_Z': [0.2, 0.2], 'DETECTIONS_PER_IMG': 200,
            'LABEL_AUG_THICKNESS_Y_TAR_ANC': [0.4, 0.4], 'LABEL_AUG_THICKNESS_Z_TAR_ANC': [0.6, 0.6],
        },
        'ROI_BOX_HEAD': {
            'FEATURE_EXTRACTOR': 'FPN2MLPFeatureExtractor', 'PREDICTOR': 'FPNPredictor',
            'POOLER_RESOLUTION': [7, 7, 3], 'POOLER_SAMPLING_RATIO': 2, 'MLP_HEAD_DIM': 512,
            'CANONICAL_SIZE': 8.0, 'POOLER_SCALES_FROM_TOP': [4, 3], 'POOLER_SCALES_SPATIAL': [],
        },
        'LOSS': {'YAW_MODE': 'Diff'},
    },
    'SPARSE3D': {
        'VOXEL_SCALE': 50, 'VOXEL_FULL_SCALE': [1536, 1536, 320], 'VAL_REPS': 3, 'RESIDUAL_BLOCK': True,
        'BLOCK_REPS': 1, 'nPlaneMap': 128, 'nPlanesFront': [32, 64, 64, 128, 128, 128, 256, 256, 256, 256],
        'KERNEL': [[2, 2, 4]] * 9, 'STRIDE': [[2, 2, 2]] * 9, 'SCENE_SIZE': [],
    },
    'DATALOADER': {'NUM_WORKERS': 4, 'SIZE_DIVISIBILITY': 0},
    'SOLVER': {
        'MAX_ITER': 40000, 'BASE_LR': 0.001, 'BIAS_LR_FACTOR': 2, 'MOMENTUM': 0.9, 'BN_MOMENTUM': 0.95,
        'TRACK_RUNNING_STATS': True, 'WEIGHT_DECAY': 0.0005, 'WEIGHT_DECAY_BIAS': 0, 'GAMMA': 0.1,
        'LR_STEP_EPOCHS': [30], 'WARMUP_FACTOR': 1.0 / 3, 'WARMUP_EPOCHS': 0.5, 'WARMUP_METHOD': 'linear',
        'CHECKPOINT_PERIOD_EPOCHS': 20, 'IMS_PER_BATCH': 16, 'EPOCHS': 100, 'EPOCHS_BETWEEN_TEST': 10,
    },
    'TEST': {'IMS_PER_BATCH': 8, 'IOU_THRESHOLD': 0.2, 'EVAL_AUG_THICKNESS_Y_TAR_ANC': [0.2, 0.2],
             'EVAL_AUG_THICKNESS_Z_TAR_ANC': [0.2, 0.2]},
    'OUTPUT_DIR': './RES',
}

_K8 = [[2, 2, 2]] * 8
_COMMON = {
    'MODEL': {'BACKBONE': {'OUT_CHANNELS': 128},
              'RPN': {'YAWS': [0, -1.57, -0.785, 0.785], 'RATIOS': [[1, 1, 1], [1, 2, 1], [2, 1, 1], [1.7, 1.7, 1]],
                      'YAW_THRESHOLD': 0.7, 'FG_IOU_THRESHOLD': 0.55, 'BG_IOU_THRESHOLD': 0.2},
              'ROI_BOX_HEAD': {'POOLER_RESOLUTION': [6, 8, 4], 'POOLER_SCALES_FROM_TOP': [4, 3]},
              'LOSS': {'YAW_MODE': 'Diff'}},
    'SPARSE3D': {'nPlanesFront': [32, 64, 64, 128, 128, 128, 256, 256, 256], 'KERNEL': _K8, 'STRIDE': _K8,
                 'VOXEL_FULL_SCALE': [4096, 4096, 512], 'VOXEL_SCALE': 50},
    'DATALOADER': {'SIZE_DIVISIBILITY': 6},
    'SOLVER': {'IMS_PER_BATCH': 1, 'BASE_LR': 0.005, 'WEIGHT_DECAY': 0.0, 'TRACK_RUNNING_STATS': False},
    'TEST': {'IMS_PER_BATCH': 1},
}
_RPN_6C = {'ANCHOR_SIZES_3D': [[0.4, 1.5, 1.5], [1.5, 1.5, 1.0], [4, 4, 1.5], [0.2, 0.5, 3], [0.4, 1.5, 3], [0.6, 2.5, 3]],
           'USE_YAWS': [1, 0, 0, 1, 1, 1], 'RPN_SCALES_FROM_TOP': [4, 3, 2, 1], 'RPN_3D_2D_SELECTOR': [1, 2, 3, 4, 5, 6]}
# values of configs/4c/4c_Fpn432_bs1_lr5_SD.yaml, configs/6c/6c_Fpn4321_bs1_lr5.yaml and
# configs/3G6c/3G6c_Fpn4321_bs1_lr2.yaml of the reference
NAMED = {
    '4c_Fpn432': [_COMMON, {
        'INPUT': {'CLASSES': ['background', 'wall', 'door', 'window']},
        'MODEL': {'RPN': {'ANCHOR_SIZES_3D': [[0.4, 1.5, 1.5], [0.2, 0.5, 3], [0.4, 1.5, 3], [0.6, 2.5, 3]],
                          'USE_YAWS': [1, 1, 1, 1], 'RPN_SCALES_FROM_TOP': [4, 3, 2], 'RPN_3D_2D_SELECTOR': [1, 3, 4, 5]}},
        'SOLVER': {'EPOCHS': 200, 'LR_STEP_EPOCHS': [100], 'WARMUP_EPOCHS': 1.0},
        'TEST': {'IOU_THRESHOLD': 0.2}, 'OUTPUT_DIR': 'RES/res_4c_Fpn432_bs1_lr5'}],
    '6c_Fpn4321': [_COMMON, {
        'INPUT': {'CLASSES': ['background', 'wall', 'door', 'window', 'ceiling', 'floor']},
        'MODEL': {'RPN': _RPN_6C},
        'SOLVER': {'EPOCHS': 20, 'LR_STEP_EPOCHS': [5], 'WARMUP_EPOCHS': 0.1},
        'TEST': {'IOU_THRESHOLD': 0.3}, 'OUTPUT_DIR': 'RES/res_6c_Fpn4321_bs1_lr5'}],
    '3G6c_Fpn4321': [_COMMON, {
        'INPUT': {'CLASSES': ['background', 'wall', 'door', 'window', 'ceiling', 'floor']},
        'MODEL': {'SEPARATE_CLASSES': [['wall'], ['ceiling', 'floor']], 'RPN': _RPN_6C},
        'SOLVER': {'EPOCHS': 20, 'LR_STEP_EPOCHS': [10], 'WARMUP_EPOCHS': 0.1, 'BASE_LR': 0.002},
        'TEST': {'IOU_THRESHOLD': 0.3}, 'OUTPUT_DIR': 'RES/res_3G6c_Fpn4321_bs1_lr2'}],
}


def class_to_label(classes):
    """data3d/suncg_utils/suncg_metas.py:14-31: ids follow the fixed CLASS_ORDER, not the yaml order."""
    assert 'background' in classes
    out, l = {}, 0
    for c in CLASS_ORDER:
        if c in classes:
            out[c] = l
            l += 1
    return out


def intact_cfg(cfg):
    """tools/train_net_sparse3d.py:231-323."""
    cfg.SPARSE3D.SCENE_SIZE = (np.array(cfg.SPARSE3D.VOXEL_FULL_SCALE, dtype=np.float64) / cfg.SPARSE3D.VOXEL_SCALE).tolist()
    rpn = cfg.MODEL.RPN
    strides = cfg.SPARSE3D.STRIDE
    n_scales = len(cfg.SPARSE3D.nPlanesFront)
    assert len(rpn.ANCHOR_SIZES_3D) == len(rpn.RPN_3D_2D_SELECTOR) == len(rpn.USE_YAWS)
    assert len(rpn.YAWS) == len(rpn.RATIOS)
    assert n_scales == len(strides) + 1
    cum = [np.array([1, 1, 1])]
    for s in range(n_scales - 1):
        cum.append(cum[-1] * np.array(strides[s]))
    per_scale = [cum[-i - 1] for i in rpn.RPN_SCALES_FROM_TOP]
    per_scale = per_scale + per_scale                                  # 3-D maps, then their 2-D projections
    rpn.ANCHOR_STRIDE = [per_scale[i].tolist() for i in rpn.RPN_3D_2D_SELECTOR]
    cs = np.cumprod(np.array(strides), 0)
    flipped = np.flip(cs, 0)
    rpn_strides = flipped[rpn.RPN_SCALES_FROM_TOP]
    rpn.RPN_MAP_SIZES = (np.array(cfg.SPARSE3D.VOXEL_FULL_SCALE).reshape(1, -1) / rpn_strides).astype(np.int32).tolist()
    spatial = np.flip(1.0 / cs, 0)[list(cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES_FROM_TOP), :]
    assert np.all(spatial[:, 0] == spatial[:, 1])
    cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES_SPATIAL = spatial[:, 0].tolist()
    c2l = class_to_label(cfg.INPUT.CLASSES)
    cfg.MODEL.SEPARATE_CLASSES_ID = [[c2l[c] for c in cs_] for cs_ in cfg.MODEL.SEPARATE_CLASSES]
    if sum(len(g) for g in cfg.MODEL.SEPARATE_CLASSES_ID) > 0:
        sep_r = 1.5 / (len(cfg.MODEL.SEPARATE_CLASSES_ID) + 1)
        for k in ('FPN_PRE_NMS_TOP_N_TRAIN', 'FPN_PRE_NMS_TOP_N_TEST', 'FPN_POST_NMS_TOP_N_TRAIN', 'FPN_POST_NMS_TOP_N_TEST'):
            rpn[k] = int(sep_r * rpn[k])
        cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = int(sep_r * cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE)
        cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG = int(sep_r * cfg.MODEL.ROI_HEADS.DETECTIONS_PER_IMG)
    return cfg


def get_cfg(name_or_path=None, opts=None):
    """Defaults -> named config or yaml file -> `opts` list -> intact_cfg (same order as
    tools/train_net_sparse3d.py:179-182)."""
    cfg = CfgNode(copy.deepcopy(_DEFAULTS))
    if name_or_path in NAMED:
        for d in NAMED[name_or_path]:
            cfg.merge_from_dict(copy.deepcopy(d))
    elif name_or_path:
        cfg.merge_from_file(name_or_path)
    if opts:
        cfg.merge_from_list(opts)
    return intact_cfg(cfg)
