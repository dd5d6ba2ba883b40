"""`seg_infer` command line -- same flags as the reference's segmentation3d/seg_infer.py:41-47
(-i input, -m model folder, -o output folder, -n seg name, -g gpu id, --save_image, --save_prob)."""
import argparse

from segmentation3d.core.seg_infer import segmentation


def main():
    parser = argparse.ArgumentParser(
        description='Sliding-window 3D medical image segmentation on an MI355X (HIP engine). Input: one MetaImage '
                    'volume (.mha/.mhd) or a text file listing volumes.')
    parser.add_argument('-i', '--input', required=True, help='input image file or list file')
    parser.add_argument('-m', '--model', required=True, help='model root folder (infer_config.py + coarse/ + fine/)')
    parser.add_argument('-o', '--output', required=True, help='output folder for segmentation')
    parser.add_argument('-n', '--seg_name', default='seg.mha', help='file name of the saved mask')
    parser.add_argument('-g', '--gpu_id', type=int, default=0, help='GPU to run on (>= 0; the engine has no CPU path)')
    parser.add_argument('--save_image', action='store_true', help='also save the input image')
    parser.add_argument('--save_prob', action='store_true', help='also save every class probability map')
    parser.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16'],
                        help='fp32 = the reference arithmetic (default); bf16 = bf16 activations / packed weights with '
                             'fp32 accumulation (about 3x faster, probabilities within ~1e-2)')
    args = parser.parse_args()
    from segmentation3d import _ops
    _ops.set_activation_dtype(args.dtype)
    segmentation(args.input, args.model, args.output, args.seg_name, args.gpu_id, False, True, args.save_image,
                 args.save_prob)


if __name__ == '__main__':
    main()
