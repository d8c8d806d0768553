"""MI355X-native force evaluation + Langevin step for timemachine."""
import os as _os

# Several contexts of one process can be stepped together on one GPU (lib.custom_ops.multiple_steps_group: windows or HREX
# replicas that share a device); each needs a hardware queue of its own -- streams that share one serialise.  The HIP runtime
# creates 4 per process by default and the null stream takes one; ask for 8 unless the user chose a number (read by the runtime when
# it first touches the device, so this has to happen before that).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
