/* vgen_host.h — C ABI of libvgen_host.so: HOST-side (x86-64, g++) helper of the pack-time weight calibration.
 *
 * Not part of the sampling hot path and not a device fallback: vgen_amd/calibrate.py chooses the 16-bit rounding of every
 * packed weight ONCE, when a model is packed (like weight loading); the denoise steps then run the single-pass launches of
 * include/vgen_hip.h on the calibrated matrices.  The reference has no counterpart (it rounds its weights to nearest under
 * amp.autocast: tools/inferences/inference_text2video_entrance.py:197, tools/modules/config.py:93); the algorithm is the
 * error-feedback rounding of GPTQ (Frantar et al. 2022, algorithm 1), restated in vgen_amd/calibrate.py.
 *
 * Built by vgen_amd/build.py::build_host from vgen_amd/csrc/host_round.cpp; bound with ctypes in
 * vgen_amd/calibrate.py::host_lib.  Plain pointers and sizes; all matrices row-major fp32 in host memory.
 */
#ifndef VGEN_HOST_H
#define VGEN_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 1 for the signatures below */
int vgen_host_abi_version(void);

/* One block [i1, i2) (at most 128 columns) of the column-by-column rounding, all N rows, rows split over `nthreads` threads.
 *   W [N, ldw]  working weight; columns [i1, i2) are read (they carry the fed-back error of the blocks before)
 *   U [K, ldu]  upper-triangular factor of the inverse of H = A^T A + damping:  H^-1 = U^T U
 *   Q [N, ldq]  out, columns [i1, i2): the chosen 16-bit values, as fp32 (exactly representable in the 16-bit type)
 *   E [N, lde]  out, column c: (w - q) / U[i1 + c, i1 + c] — the caller applies W[:, i2:] -= E . U[i1:i2, i2:]
 *   dtype       0 = fp16, 1 = bf16 (round to nearest even)
 * Returns 0; -1 if the block is empty or wider than 128, or dtype is unknown.  Same arithmetic, bit for bit, as
 * vgen_amd/calibrate.py::_round_block_torch (tests/test_calibrate.py). */
int vgen_host_gptq_block(const float *W, int64_t N, int64_t ldw, const float *U, int64_t ldu, int64_t i1, int64_t i2,
                         int dtype, float *Q, int64_t ldq, float *E, int64_t lde, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
