"""One-image tool with the reference's argv (test_FCN_binarizer.py:13-59): config network input_img output_prefix ->
<prefix>_BIN.png, _text.png, _bg.png.  No inversion here (unlike the step-01 worker)."""
import sys


def main():
    if len(sys.argv) < 5:
        print("Usage\n\tpython {0:s} config network input_img output_prefix".format(sys.argv[0]))
        return
    import numpy as np
    import PIL.Image
    import torch
    from AM_CommonTools.configuration.configuration import Configuration
    from AccessMath.lecturenet_v1.FCN_lecturenet import FCN_LectureNet
    config = Configuration.from_file(sys.argv[1])
    net = FCN_LectureNet.CreateFromConfig(config, 3, False)
    net.load_state_dict(torch.load(sys.argv[2], map_location="cpu"))
    net.eval()
    net = net.cuda()
    pil = PIL.Image.open(sys.argv[3]).convert("RGB")
    binary, text_mask, rec_img = net.binarize(pil, return_others=True, force_binary=True)
    PIL.Image.fromarray(binary).save(sys.argv[4] + "_BIN.png")
    PIL.Image.fromarray(text_mask).save(sys.argv[4] + "_text.png")
    PIL.Image.fromarray(np.ascontiguousarray(rec_img[:, :, ::-1])).save(sys.argv[4] + "_bg.png")


if __name__ == "__main__":
    main()
