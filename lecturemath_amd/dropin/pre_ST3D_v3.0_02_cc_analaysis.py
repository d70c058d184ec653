"""Step 02 entry point (file name kept as the reference spells it): binary frames -> CC labelling + temporal CC matching on
the MI355X -> (frame_times, frame_indices, estimator), pickled by the harness as <CC_STABILITY_OUTPUT><lecture>.dat."""
import sys


def process_input(process, input_data):
    from AccessMath.preprocessing.content.helper import Helper
    from AccessMath.preprocessing.content.cc_stability_estimator import CCStabilityEstimator
    frame_times, frame_indices, compressed_frames = input_data
    print("Decompressing input...")
    binary_frames = Helper.decompress_binary_images(compressed_frames)
    height, width = binary_frames[0].shape
    cfg = process.configuration
    estimator = CCStabilityEstimator(width, height, cfg.get_float("CC_STABILITY_MIN_RECALL", 0.925),
                                     cfg.get_float("CC_STABILITY_MIN_PRECISION", 0.925), cfg.get_int("CC_STABILITY_MAX_GAP", 85), True)
    print("Processing frames...")
    for frame in binary_frames:
        estimator.add_frame(frame, True)
    estimator.finish_processing()
    return frame_times, frame_indices, estimator


def main():
    from AccessMath.preprocessing.user_interface.console_ui_process import ConsoleUIProcess
    if not ConsoleUIProcess.usage_with_config_check(sys.argv):
        return
    process = ConsoleUIProcess.FromConfigPath(sys.argv[1], sys.argv[2:], "BINARIZATION_OUTPUT", "CC_STABILITY_OUTPUT")
    if not process.initialize():
        return
    process.start_input_processing(process_input)
    print("Finished!")


if __name__ == "__main__":
    main()
