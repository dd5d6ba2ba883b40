"""Default inference configuration -- same sections and keys as the reference's config/infer_config.py:9-113
(general / coarse / fine).  Copied next to trained models by `seg_train` and read back by `load_models`."""
from easydict import EasyDict as edict

__C = edict()
cfg = __C

__C.general = {}
__C.general.single_scale = 'DISABLE'                   # 'coarse' | 'fine' | 'DISABLE' (= coarse-to-fine cascade)

__C.coarse = {}
__C.coarse.model_name = 'coarse'
__C.coarse.pick_largest_cc = True
__C.coarse.remove_small_cc = 0
__C.coarse.partition_type = 'DISABLE'                  # whole volume in one pass
__C.coarse.partition_size = [51.2, 51.2, 51.2]         # mm
__C.coarse.partition_stride = [51.2, 51.2, 51.2]
__C.coarse.cpu_model_spacing_increase_ratio = 1.0
__C.coarse.cpu_partition_decrease_ratio = 1.0

__C.fine = {}
__C.fine.model_name = 'fine'
__C.fine.pick_largest_cc = True
__C.fine.remove_small_cc = 0
__C.fine.partition_type = 'SIZE'                       # sliding window
__C.fine.partition_size = [96.0, 96.0, 96.0]           # mm
__C.fine.partition_stride = [48.0, 48.0, 48.0]
__C.fine.cpu_model_spacing_increase_ratio = 1.0
__C.fine.cpu_partition_decrease_ratio = 1.0
