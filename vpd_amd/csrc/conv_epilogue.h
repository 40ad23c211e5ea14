// Epilogue shared by the convolution kernels (bf16 NHWC store, eval scale/shift/residual/ReLU,
// accumulate, per-channel statistics).
#pragma once
#include "common.h"

// Shared epilogue: acc[a][b][j] = out[channel n0 + wn*WTN + a*16 + 4*fq + j][pixel m0 + wm*WTM + b*16 + fr].
// Stores bf16 NHWC (8 B per lane), optional eval epilogue (scale/shift/residual/ReLU), optional
// read-modify-write accumulate, optional per-channel sum / sum-of-squares of the stored values.
template <int BM, int BN, int WM, int WN>
static __device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[BN / WN / 16][BM / WM / 16],
                                                     int mtile, int n0, unsigned char* smem) {
    constexpr int WTM = BM / WM;
    constexpr int WTN = BN / WN;
    constexpr int MI = WTM / 16;
    constexpr int NI = WTN / 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int fr = lane & 15;
    const int fq = lane >> 4;
    const int m0 = mtile * BM;
    const int HW = p.Hs * p.Ws;
    // acc[a][b][j] = out[channel n0 + wn*WTN + a*16 + 4*fq + j][pixel m0 + wm*WTM + b*16 + fr]
    float s1[NI][4], s2[NI][4];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[a][j] = 0.f; s2[a][j] = 0.f; }

#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = m0 + wm * WTM + b * 16 + fr;
        const bool valid = m < p.M;
        const int mc = valid ? m : p.M - 1;
        const int bi = mc / HW;
        const int r = mc - bi * HW;
        const int yy = r / p.Ws;
        const int xx = r - yy * p.Ws;
        const size_t yoff = ((size_t)(bi * p.yHp + yy * p.osub + p.oph + p.ypad) * p.yWp +
                             (xx * p.osub + p.opw + p.ypad)) * p.yC;
        size_t roff = 0;
        if (p.res) roff = ((size_t)(bi * p.rHp + yy + p.rpad) * p.rWp + (xx + p.rpad)) * p.rC;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int n = n0 + wn * WTN + a * 16 + 4 * fq;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (p.ep_scale) {
                const float4 sc = *reinterpret_cast<const float4*>(p.ep_scale + n);
                const float4 sh = *reinterpret_cast<const float4*>(p.ep_shift + n);
                v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y;
                v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w;
                if (p.res) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(p.res + roff + n);
                    v[0] += bf2f((unsigned short)(rv.x & 0xffff)); v[1] += bf2f((unsigned short)(rv.x >> 16));
                    v[2] += bf2f((unsigned short)(rv.y & 0xffff)); v[3] += bf2f((unsigned short)(rv.y >> 16));
                }
                if (p.ep_relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                }
            }
            bf16_t* dst = p.y + yoff + n;
            if (p.accumulate && valid) {
                const uint2 ov = *reinterpret_cast<const uint2*>(dst);
                v[0] += bf2f((unsigned short)(ov.x & 0xffff)); v[1] += bf2f((unsigned short)(ov.x >> 16));
                v[2] += bf2f((unsigned short)(ov.y & 0xffff)); v[3] += bf2f((unsigned short)(ov.y >> 16));
            }
            uint2 ov;
            ov.x = pack2bf(v[0], v[1]);
            ov.y = pack2bf(v[2], v[3]);
            if (valid) {
                *reinterpret_cast<uint2*>(dst) = ov;
                // statistics are taken over the bf16-rounded values actually stored
                const float q0 = bf2f((unsigned short)(ov.x & 0xffff)), q1 = bf2f((unsigned short)(ov.x >> 16));
                const float q2 = bf2f((unsigned short)(ov.y & 0xffff)), q3 = bf2f((unsigned short)(ov.y >> 16));
                s1[a][0] += q0; s2[a][0] += q0 * q0;
                s1[a][1] += q1; s2[a][1] += q1 * q1;
                s1[a][2] += q2; s2[a][2] += q2 * q2;
                s1[a][3] += q3; s2[a][3] += q3 * q3;
            }
        }
    }

    if (p.stats) {
        // reduce over the 16 pixel lanes, then over the WM pixel-waves through LDS
        float* red = reinterpret_cast<float*>(smem);      // [WM][2][BN] (staging LDS is free now)
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float u = s1[a][j], v = s2[a][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    u += __shfl_xor(u, o, 64);
                    v += __shfl_xor(v, o, 64);
                }
                if (fr == 0) {
                    const int c = wn * WTN + a * 16 + 4 * fq + j;
                    red[(wm * 2 + 0) * BN + c] = u;
                    red[(wm * 2 + 1) * BN + c] = v;
                }
            }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN;
            const int c = tid - which * BN;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + c];
            // VPD_STAT_ROWS accumulator rows spread the atomic traffic; bn_finalize sums and re-zeroes them
            atomicAdd(&p.stats[((size_t)(mtile & (VPD_STAT_ROWS - 1)) * 2 + which) * p.Co + n0 + c], t);
        }
    }
}

