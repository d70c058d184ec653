"""Step 05 entry point (same name, argv, config keys, inputs and output as the reference's
pre_ST3D_v3.0_05_generate_summary.py:17-73): [SpaceTimeStruct, video segments] -> keyframes per segment, exported with the
reference's own KeyframeExporter (XML + PNG files; not part of this build, imported from the reference tree)."""
import sys


def tiling_intervals(st3D, video_segments):
    """(idx_intervals, time_intervals, summary_indices, summary_times) of pre_ST3D_v3.0_05_generate_summary.py:38-66."""
    # Every segment is summarised at its last frame (:63-66); the exported intervals tile the video: each boundary sits in the
    # middle of the gap between one segment's last frame and the next segment's first frame (:46-57), the first interval
    # starts at 0 and the last ends with the last segment.
    summary_indices = [st3D.frame_indices[last] for _, last in video_segments]
    summary_times = [st3D.frame_times[last] for _, last in video_segments]
    following = [(st3D.frame_indices[first], st3D.frame_times[first]) for first, _ in video_segments[1:]]
    cut_idx = [int((end + nxt[0]) / 2) for end, nxt in zip(summary_indices, following)] + summary_indices[-1:]
    cut_time = [(end + nxt[1]) / 2.0 for end, nxt in zip(summary_times, following)] + summary_times[-1:]
    idx_intervals = list(zip([0] + cut_idx[:-1], cut_idx))
    time_intervals = list(zip([0] + cut_time[:-1], cut_time))
    return idx_intervals, time_intervals, summary_indices, summary_times


def process_input(process, input_data):
    from AccessMath.preprocessing.content.keyframe_extractor import KeyframeExtractor
    st3D = input_data[0]
    video_segments = input_data[1]
    keyframes, cc_times = KeyframeExtractor.GenerateFromST3DForIntervals(st3D, video_segments)
    idx_intervals, time_intervals, summary_indices, summary_times = tiling_intervals(st3D, video_segments)
    if getattr(process, "database", None) is not None:
        from AccessMath.preprocessing.content.keyframe_exporter import KeyframeExporter
        database, lecture = process.database, process.current_lecture
        output_prefix = process.configuration.get("OUTPUT_PATH") + "/" + database.output_summaries + "/" + database.name + "_" + lecture.title.lower()
        print("Saving data to: " + output_prefix)
        KeyframeExporter.Export(output_prefix, database, lecture, idx_intervals, time_intervals, summary_indices, summary_times, keyframes)
        KeyframeExporter.ExportGUIInfo(output_prefix, cc_times)
    return (summary_indices, summary_times, keyframes),


def main():
    import lm_entry
    lm_entry.run_on_inputs(sys.argv, ["CC_ST3D_OUTPUT", "VIDEO_SEGMENTATION_OUTPUT"], "SUMMARY_KEYFRAMES_OUTPUT", process_input)


if __name__ == "__main__":
    main()
