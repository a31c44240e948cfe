"""MI355X-native hot path of LLMRec Stage 2: the C-ABI HIP library (csrc/, lib/) and its Python host side."""
import os
import sys
import warnings

# ROCm's hipGraphLaunch gives a multi-branch graph's internal streams to its branches and skips every internal stream that shares a HARDWARE
# queue with the launch stream - without a bounds check (libamdhip64 of this image; DESIGN.md section 4, "HIP graph launch and hardware queues").
# With the default pool of 4 hardware queues two of an executable's four internal streams can land on the launch stream's queue, and the
# launch then walks off the stream vector (a host-side segfault that depended on every stream the process had created before). Eight queues
# keep the four internal streams on distinct queues. Must be set before the HIP runtime initialises (first CUDA call of the process).
#
# The setting only works if it is in the environment BEFORE the runtime initialises. An embedder that touched the GPU first (a notebook, a
# host program patched as INTEGRATION.md section B shows, a torch.distributed worker) - or that exported a smaller value - would get the
# fault back with no diagnostic. So the decision is made here, once, and it is loud: graph replay is REFUSED in that process
# (FusedStep.capture raises, main.py / bench.py fall back to eager launches with a warning) unless LLMREC_UNSAFE_GRAPH=1 overrides.
MIN_HW_QUEUES = 8


def queue_decision(env_value, hip_initialised: bool):
    """Pure function of (the GPU_MAX_HW_QUEUES value in the environment or None, whether the HIP runtime is already up) ->
    (action, graph_replay_safe, message): action "set" = put 8 into the environment now; "keep" = leave the environment alone."""
    if env_value is None:
        if not hip_initialised:
            return "set", True, None
        return "keep", False, ("llmrec_amd was imported after the HIP runtime initialised and GPU_MAX_HW_QUEUES was not set: the runtime keeps its default "
                               "of 4 hardware queues, with which hipGraphLaunch of the multi-stream step graph can fault on the host (DESIGN.md section 4). "
                               "HIP-graph replay is disabled in this process; export GPU_MAX_HW_QUEUES=8 (or import llmrec_amd before the first CUDA call) "
                               "to enable it")
    try:
        n = int(env_value)
    except ValueError:
        n = -1
    if n >= MIN_HW_QUEUES:
        return "keep", True, None
    return "keep", False, ("GPU_MAX_HW_QUEUES=%s is below %d: hipGraphLaunch of the multi-stream step graph can fault on the host with fewer hardware queues "
                           "(DESIGN.md section 4). HIP-graph replay is disabled in this process; export GPU_MAX_HW_QUEUES=%d to enable it"
                           % (env_value, MIN_HW_QUEUES, MIN_HW_QUEUES))


def _hip_initialised() -> bool:
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


_action, _graph_safe, _message = queue_decision(os.environ.get("GPU_MAX_HW_QUEUES"), _hip_initialised())
if _action == "set":
    os.environ["GPU_MAX_HW_QUEUES"] = str(MIN_HW_QUEUES)
if _message is not None:
    warnings.warn(_message, RuntimeWarning, stacklevel=2)


def graph_replay_safe() -> bool:
    """False when this process cannot rely on the hardware-queue work-around (see above); LLMREC_UNSAFE_GRAPH=1 overrides."""
    return _graph_safe or os.environ.get("LLMREC_UNSAFE_GRAPH", "0") == "1"


def require_graph_replay(what: str):
    """Raise unless HIP-graph replay is safe in this process (called by every capture of a multi-stream graph)."""
    if not graph_replay_safe():
        raise RuntimeError("%s: HIP-graph replay refused - %s" % (what, _message))
