// librmd_hip.so as ONE translation unit: A/B builds with the retired matchers (-DRMD_AB_MATCHERS, tools/ab_make.sh), whose headers define
// kernels that must not be compiled twice.  The product library is built from the separate units (rpg_open_remode_amd/build.py).
#include "rmd_capi.hip"
#include "rmd_update.hip"
#include "rmd_ingest.hip"
#include "rmd_batch.hip"
#include "rmd_denoise.hip"
#include "rmd_reduce.hip"
