"""Recurrent bottlenecks of the stage U-Nets (BOTTLENECK=CLSTM|CGRU).

The reference imports `ConvBLSTM` / `ConvBGRU` from this package path
(scripts/models/flow_computation.py:7-8), a git submodule that is empty in the
reference tree (.gitmodules:1-3 -> SreenivasVRao/ConvGRU-ConvLSTM-PyTorch).  The
modules here keep that package's constructor arguments, parameter names and
`forward(x_fwd, x_rev)` contract as used at flow_computation.py:73-88,208-211 and
run on the HIP kernels (ssm_amd.engine.RecurrentBottleneck).  Their arithmetic
restates the package's published cells; parity with the original is UNPINNED
(see oracle/ssm_oracle.py and DESIGN.md).
"""
