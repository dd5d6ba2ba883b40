"""Default training configuration -- same sections and keys as the reference's config/train_config.py:13-140
(general / dataset / loss / net / train / debug), so existing config files can be passed to `seg_train -i` unchanged."""
from easydict import EasyDict as edict
from segmentation3d.utils.normalizer import FixedNormalizer, AdaptiveNormalizer  # noqa: F401

__C = edict()
cfg = __C

__C.general = {}
__C.general.imseg_list = '/path/to/train.csv'          # image / segmentation pair list
__C.general.save_dir = '/tmp/seg3d_model'              # models, logs and checkpoints go here
__C.general.model_scale = 'fine'                       # sub-folder of save_dir ('coarse' / 'fine' / ...)
__C.general.resume_epoch = -1                          # >= 0: continue from that checkpoint
__C.general.num_gpus = 1                               # > 0 (one process per GPU; the HIP engine has no CPU path)
__C.general.seed = 0

__C.dataset = {}
__C.dataset.num_classes = 2
__C.dataset.spacing = [1.0, 1.0, 1.0]                  # mm, (x, y, z)
__C.dataset.crop_size = [96, 96, 96]                   # voxels, multiples of max_stride (16)
__C.dataset.sampling_method = 'HYBRID'                 # GLOBAL | MASK | HYBRID | CENTER
__C.dataset.random_translation = [15, 15, 15]             # mm, uniform in [-t, t] per axis
__C.dataset.random_scale = [0.9, 1.1]                     # crop spacing = spacing * uniform(lo, hi)
__C.dataset.interpolation = 'LINEAR'                   # NN | LINEAR
__C.dataset.crop_normalizers = [AdaptiveNormalizer()]

__C.loss = {}
__C.loss.name = 'Dice'                                 # Focal | Dice | CE
__C.loss.obj_weight = [1 / 2, 1 / 2]                   # per-class weights (normalised to sum 1)
__C.loss.focal_gamma = 2

__C.net = {}
__C.net.name = 'vbnet'                                 # plugin module under segmentation3d.network

__C.train = {}
__C.train.epochs = 1001
__C.train.batchsize = 4                                # per GPU
__C.train.num_threads = 4
__C.train.lr = 1e-4
__C.train.betas = (0.9, 0.999)
__C.train.save_epochs = 100
# extensions of this build (absent keys mean the reference's behaviour: fp32, eager steps)
__C.train.compute_dtype = 'fp32'                       # 'bf16': bf16 activations / packed weights, fp32 accumulate + master weights
__C.train.use_graph = False                            # capture the whole train step in one hipGraph (single process only)
__C.train.gc_freeze = True                             # gc.freeze() after the first steps: short collector pauses in the eager loop

__C.debug = {}
__C.debug.save_inputs = False
