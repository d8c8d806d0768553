"""MI355X-native force evaluation + Langevin step for timemachine."""
# (GPU_MAX_HW_QUEUES -- the hardware queues that custom_ops.multiple_steps_group's streams need -- is exported by the native
# library itself when it is loaded: timemachine_amd/csrc/c_api.cpp, include/timemachine_amd.h.)
