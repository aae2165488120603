"""MI355X-native Super SloMo interpolation path: host side above the C-ABI.

`hipbind`  ctypes binding of csrc/libssm_hip.so (include/ssm_hip.h)
`engine`   plan/arena/launch sequence for frame pair -> intermediate frames
`weights`  layer table + deterministic synthetic weights/frames
"""
