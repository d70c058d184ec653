"""Step 02 entry point (file name kept as the reference spells it): binary frames -> CC labelling + temporal CC matching on
the MI355X -> (frame_times, frame_indices, estimator), pickled by the harness as <CC_STABILITY_OUTPUT><lecture>.dat."""
import sys


def process_input(process, input_data):
    from AccessMath.preprocessing.content.helper import Helper
    from AccessMath.preprocessing.content.cc_stability_estimator import CCStabilityEstimator
    times, indices, png_frames = input_data
    cfg = process.configuration
    thresholds = [cfg.get_float("CC_STABILITY_MIN_" + which, 0.925) for which in ("RECALL", "PRECISION")]
    print("Decompressing input...")
    frames = Helper.decompress_binary_images(png_frames)
    estimator = CCStabilityEstimator(frames[0].shape[1], frames[0].shape[0], thresholds[0], thresholds[1],
                                     cfg.get_int("CC_STABILITY_MAX_GAP", 85), True)
    print("Processing frames...")
    for binary in frames:                       # buffered; pushed to the device a batch at a time
        estimator.add_frame(binary, True)
    estimator.finish_processing()
    return times, indices, estimator


def main():
    import lm_entry
    lm_entry.run_on_inputs(sys.argv, "BINARIZATION_OUTPUT", "CC_STABILITY_OUTPUT", process_input)


if __name__ == "__main__":
    main()
