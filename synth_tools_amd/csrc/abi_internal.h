// abi_internal.h -- private helpers shared by the abi_*.cpp files of libsynth_mi355x.so.
#pragma once
#include "smx_common.h"
#include <vector>

using smx::set_error;

// linux/synth.c:180: (1.0 / 2^32) * (float)sum, product in double, result float
static inline float bus_to_float(int32_t sum)
{
    return (float)((1.0 / 4294967296.0) * (double)(float)sum);
}


// abi_saw.cpp: struct synth's 64 voices for one block (the drop-in synth_run's device path).  Not exported.
namespace smx { int bank_dropin_run(smx_bank *b, const uint32_t *inc, const uint32_t *state, float *vec, int n); }

// Grow-only device scratch buffer (waits for the stream before reallocating).
static inline int dev_reserve(void **ptr, size_t *cap, size_t need, hipStream_t stream)
{
    if (need <= *cap) return SMX_OK;
    SMX_HIP(hipStreamSynchronize(stream));
    if (*ptr) SMX_HIP(hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    SMX_HIP(hipMalloc(ptr, need));
    *cap = need;
    return SMX_OK;
}


// The oscillator bank's state is also read by the firmware control surface (smx_fw_poll).
struct smx_osc {
    uint32_t n = 0, n_pad = 0, log_max = 26;
    int device = 0;
    uint32_t *d_phase = nullptr, *d_speed = nullptr;
    smx::PmeasArrays pm{};
    void *d_tmp = nullptr; size_t tmp_cap = 0;      // sync/valid bits or timestamps
    void *d_tmp2 = nullptr; size_t tmp2_cap = 0;
    uint8_t *d_duty = nullptr; size_t duty_cap = 0;
    hipStream_t stream = nullptr;
};

