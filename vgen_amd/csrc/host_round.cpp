// libvgen_host.so — HOST code (g++, no device part): the column loop of the calibrated rounding (vgen_amd/calibrate.py).
//
// Pack-time work, never inside a denoise step.  GPTQ-style error feedback (Frantar et al. 2022, algorithm 1) rounds one
// K-column at a time and spreads its rounding error over the columns not yet rounded; ROWS of the weight are independent,
// so a block of columns is processed row by row — a row's block (<= 128 floats) and the block's triangle of U stay in
// cache — and the rows are split over threads.  The Python loop this replaces issued ~6 tiny tensor ops per column
// (0.8 M columns for the 1.4 G-parameter UNet: 205 of the pass's 267 s on 8 cores).
//
// Arithmetic is kept IDENTICAL to the torch loop (vgen_amd/calibrate.py::_round_block_torch, the tested restatement):
// fp32 throughout, one rounding per operation (-ffp-contract=off: no fused multiply-add), IEEE division, round-to-nearest-
// even conversions (F16C for fp16 — what torch's CPU conversion does — and the usual integer trick for bf16), so the two
// agree bit for bit (tests/test_calibrate.py).
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
#include <immintrin.h>

namespace {

inline float round_fp16(float x) { return _cvtsh_ss(_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }

inline float round_bf16(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) {            // NaN stays NaN
        u = 0x7fc00000u;
    } else {
        u += 0x7fffu + ((u >> 16) & 1u);
        u &= 0xffff0000u;
    }
    float y;
    std::memcpy(&y, &u, 4);
    return y;
}

template <bool BF16>
void rows(const float* W, int64_t ldw, const float* U, int64_t ldu, int64_t i1, int64_t bs, float* Q, int64_t ldq, float* E,
          int64_t lde, int64_t r0, int64_t r1) {
    float w[128];
    for (int64_t r = r0; r < r1; ++r) {
        const float* wr = W + r * ldw + i1;
        std::memcpy(w, wr, sizeof(float) * bs);
        float* qr = Q + r * ldq + i1;
        float* er = E + r * lde;
        for (int64_t i = 0; i < bs; ++i) {
            const float* u = U + (i1 + i) * ldu + i1;
            const float q = BF16 ? round_bf16(w[i]) : round_fp16(w[i]);
            const float e = (w[i] - q) / u[i];
            qr[i] = q;
            er[i] = e;
            for (int64_t j = i; j < bs; ++j) w[j] = w[j] - e * u[j];
        }
    }
}

}  // namespace

extern "C" {

// One column block [i1, i2) of the error-feedback rounding, all N rows.
//   W  [N, ldw] fp32 working weight: columns [i1, i2) are read (they already carry the feedback of the blocks before)
//   U  [K, ldu] fp32 upper-triangular factor, H^-1 = U^T U
//   Q  [N, ldq] fp32 out: the rounded values of columns [i1, i2) (exactly representable in the 16-bit type)
//   E  [N, lde] fp32 out: column c = (w - q) / U[c, c] of block column c — the caller's trailing update is E @ U[i1:i2, i2:]
//   dtype 0 = fp16, 1 = bf16;  returns 0, or -1 on a bad argument (block wider than 128, unknown dtype)
int vgen_host_gptq_block(const float* W, int64_t N, int64_t ldw, const float* U, int64_t ldu, int64_t i1, int64_t i2, int dtype,
                         float* Q, int64_t ldq, float* E, int64_t lde, int nthreads) {
    const int64_t bs = i2 - i1;
    if (bs <= 0 || bs > 128 || (dtype != 0 && dtype != 1) || N < 0) return -1;
    int nt = std::max(1, std::min<int>(nthreads, (int)((N + 31) / 32)));
    auto run = [&](int64_t r0, int64_t r1) {
        if (dtype == 1) rows<true>(W, ldw, U, ldu, i1, bs, Q, ldq, E, lde, r0, r1);
        else rows<false>(W, ldw, U, ldu, i1, bs, Q, ldq, E, lde, r0, r1);
    };
    if (nt == 1) {
        run(0, N);
        return 0;
    }
    std::vector<std::thread> th;
    const int64_t per = (N + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
        const int64_t r0 = t * per, r1 = std::min<int64_t>(N, r0 + per);
        if (r0 < r1) th.emplace_back(run, r0, r1);
    }
    for (auto& t : th) t.join();
    return 0;
}

int vgen_host_abi_version() { return 1; }

}  // extern "C"
