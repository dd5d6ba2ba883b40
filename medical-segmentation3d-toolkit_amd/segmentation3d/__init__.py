"""segmentation3d -- MI355X-native drop-in for the hot path of qinliuliuqin/Medical-Segmentation3d-Toolkit.

Same import paths and call signatures as the reference for the network plugins (`segmentation3d.network.vnet`,
`.vbnet`), the losses (`segmentation3d.loss.*`), the train step and the sliding-window inference path; the arithmetic
runs in hand-written HIP kernels for gfx950 (libseg3d_hip.so, C ABI in include/seg3d_hip.h).
"""
__version__ = '0.1.0'
