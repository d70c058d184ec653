"""Shared command-line wiring of the pre_ST3D_* entry points of this overlay: the harness (ConsoleUIProcess, database,
lecture selection, pickled inputs / outputs) stays the reference's; a script only names its config keys and its callbacks."""
import time


def run_on_inputs(argv, input_keys, output_keys, process_input, choose_inputs=None):
    """Scripts that consume the pickled outputs of earlier steps (02, 03, 04, 05)."""
    from AccessMath.preprocessing.user_interface.console_ui_process import ConsoleUIProcess
    if not ConsoleUIProcess.usage_with_config_check(argv):
        return
    process = ConsoleUIProcess.FromConfigPath(argv[1], argv[2:], input_keys, output_keys)
    if choose_inputs is not None:
        process.input_temp_prefix = choose_inputs(process.configuration)
    if process.initialize():
        begin = time.time()
        process.start_input_processing(process_input)
        print("Total time: %.1f s" % (time.time() - begin))
        print("Finished")


def run_on_videos(argv, output_keys, get_worker, get_results, default_fps=1.0):
    """Step 01: sampled video frames go through a worker object."""
    from AccessMath.preprocessing.user_interface.console_ui_process import ConsoleUIProcess
    if not ConsoleUIProcess.usage_with_config_check(argv):
        return
    process = ConsoleUIProcess.FromConfigPath(argv[1], argv[2:], None, output_keys)
    if process.initialize():
        process.start_video_processing(process.configuration.get_float("SAMPLING_FPS", default_fps), get_worker, get_results, 0, True, True)
        print("Finished")
