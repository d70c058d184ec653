"""Step 05 entry point (same name, argv, config keys, inputs and output as the reference's
pre_ST3D_v3.0_05_generate_summary.py:17-73): [SpaceTimeStruct, video segments] -> keyframes per segment, exported with the
reference's own KeyframeExporter (XML + PNG files; not part of this build, imported from the reference tree)."""
import sys


def process_input(process, input_data):
    from AccessMath.preprocessing.content.keyframe_extractor import KeyframeExtractor
    st3D = input_data[0]
    video_segments = input_data[1]
    keyframes, cc_times = KeyframeExtractor.GenerateFromST3DForIntervals(st3D, video_segments)
    idx_intervals, time_intervals, summary_times, summary_indices = [], [], [], []
    # logical frame indices of the intervals -> absolute frame indices; boundaries moved to the middle of the gaps (:38-66)
    last_start = 0
    last_time_start = 0
    for idx, (segment_start, segment_end) in enumerate(video_segments):
        frame_end = st3D.frame_indices[segment_end]
        time_end = st3D.frame_times[segment_end]
        if idx + 1 < len(video_segments):
            next_frame_start = st3D.frame_indices[video_segments[idx + 1][0]]
            next_time_start = st3D.frame_times[video_segments[idx + 1][0]]
            interval_end = int((frame_end + next_frame_start) / 2)
            time_interval_end = (time_end + next_time_start) / 2.0
        else:
            interval_end = frame_end
            time_interval_end = time_end
        idx_intervals.append((last_start, interval_end))
        time_intervals.append((last_time_start, time_interval_end))
        last_start = interval_end
        last_time_start = time_interval_end
        summary_indices.append(frame_end)
        summary_times.append(st3D.frame_times[segment_end])
    if getattr(process, "database", None) is not None:
        from AccessMath.preprocessing.content.keyframe_exporter import KeyframeExporter
        database, lecture = process.database, process.current_lecture
        output_prefix = process.configuration.get("OUTPUT_PATH") + "/" + database.output_summaries + "/" + database.name + "_" + lecture.title.lower()
        print("Saving data to: " + output_prefix)
        KeyframeExporter.Export(output_prefix, database, lecture, idx_intervals, time_intervals, summary_indices, summary_times, keyframes)
        KeyframeExporter.ExportGUIInfo(output_prefix, cc_times)
    return (summary_indices, summary_times, keyframes),


def main():
    from AccessMath.preprocessing.user_interface.console_ui_process import ConsoleUIProcess
    if not ConsoleUIProcess.usage_with_config_check(sys.argv):
        return
    process = ConsoleUIProcess.FromConfigPath(sys.argv[1], sys.argv[2:], ["CC_ST3D_OUTPUT", "VIDEO_SEGMENTATION_OUTPUT"],
                                              "SUMMARY_KEYFRAMES_OUTPUT")
    if not process.initialize():
        return
    process.start_input_processing(process_input)
    print("Finished")


if __name__ == "__main__":
    main()
