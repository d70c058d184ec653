"""Step 01 entry point (same name, argv and outputs as the reference's pre_ST3D_v3.0_01_binarize.py): sample the lecture
videos, binarize every sampled frame with FCN-LectureNet on the MI355X, hand (frame_times, frame_indices, compressed_frames)
to the harness, which pickles them as <BINARIZATION_OUTPUT><lecture>.dat."""
import sys


def get_worker(process):
    import torch
    from AccessMath.lecturenet_v1.FCN_lecturenet import FCN_LectureNet
    from AccessMath.preprocessing.video_worker.FCN_lecturenet_binarizer import FCN_LectureNet_Binarizer
    print("... loading model ...")
    cfg = process.configuration
    model_filename = (cfg.get_str("OUTPUT_PATH") + "/" + cfg.get_str("BINARIZATION_FCN_LECTURENET_DIR") + "/" +
                      cfg.get_str("BINARIZATION_FCN_LECTURENET_FILENAME"))
    lecture_net = FCN_LectureNet.CreateFromConfig(cfg, 3, False)
    lecture_net.load_state_dict(torch.load(model_filename, map_location="cpu"))
    lecture_net.eval()
    lecture_net = lecture_net.cuda()          # FCN_BINARIZER_USE_CUDA is moot: the HIP path is the only path
    worker = FCN_LectureNet_Binarizer(lecture_net)
    worker.set_debug_mode(cfg.get("BINARIZATION_DEBUG_MODE", False), 0, cfg.get_int("BINARIZATION_DEBUG_END_TIME", 50000),
                          getattr(process, "img_dir", None), getattr(getattr(process, "current_lecture", None), "title", ""))
    return worker


def get_results(worker):
    del worker.lecture_net
    return worker.frame_times, worker.frame_indices, worker.compressed_frames


def main():
    import lm_entry
    lm_entry.run_on_videos(sys.argv, "BINARIZATION_OUTPUT", get_worker, get_results)


if __name__ == "__main__":
    main()
