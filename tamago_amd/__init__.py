"""tamago_amd - TamaGo's batched MCTS leaf-evaluation path on MI355X.

Importing the package asks the HIP runtime for more hardware queues than its default of four, unless the process already chose
(GPU_MAX_HW_QUEUES): a self-play shard drives several HIP streams at once (sub-groups of a lock-step move, lanes on different
moves, the random-stream generator), and streams that share a hardware queue serialise - measured at 16 boards in two lanes:
2.66 M leaf-evals/s on 8 queues, 3.65 M on 16 (profiles/r06_selfplay_lanes_sweep.txt).  The runtime reads the variable when it
loads (measured: setting it after `import torch` has no effect), so it has to be in the environment before torch is imported:
HW_QUEUES records what this process will get - the variable's value, 16 if the package could still set it (torch not imported
yet), None if that is not knowable (callers that split a shard into groups or lanes can look at it).
bench.py, tests/conftest.py and the self-play launcher set it first thing.
"""
import os as _os


def _hardware_queues():
    env = _os.environ.get("GPU_MAX_HW_QUEUES")
    if env is not None:
        try:
            return int(env)
        except ValueError:
            return None
    import sys
    if "torch" in sys.modules:
        return None                                      # too late to ask: the runtime took its settings when torch loaded it
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
    return 16


HW_QUEUES = _hardware_queues()
