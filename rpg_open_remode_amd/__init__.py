"""MI355X-native REMODE depth-filter path (HIP kernels behind a C ABI) -- see DESIGN.md.

`rpg_open_remode_amd.api` mirrors the reference's C++ interface (rmd::SeedMatrix, rmd::DepthmapDenoiser,
rmd::ImageReducer, rmd::DeviceImage, rmd::Depthmap); `rpg_open_remode_amd.synth` renders the synthetic
test sequences; `rpg_open_remode_amd.build` compiles the native libraries in-tree.
"""
from .api import (ConvergenceStates, Depthmap, DepthmapDenoiser, DeviceImage, ImageReducer, PinholeCamera, RmdHipError,  # noqa: F401
                  SE3, SeedMatrix, checkCudaDevice)
