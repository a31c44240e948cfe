"""MI355X-native hot path of LLMRec Stage 2: the C-ABI HIP library (csrc/, lib/) and its Python host side."""
import os

# ROCm's hipGraphLaunch gives a multi-branch graph's internal streams to its branches and skips every internal stream that shares a HARDWARE
# queue with the launch stream - without a bounds check (libamdhip64 of this image; DESIGN.md section 4, "HIP graph launch and hardware queues").
# With the default pool of 4 hardware queues two of an executable's four internal streams can land on the launch stream's queue, and the
# launch then walks off the stream vector (a host-side segfault that depended on every stream the process had created before). Eight queues
# keep the four internal streams on distinct queues. Must be set before the HIP runtime initialises (first CUDA call of the process).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
