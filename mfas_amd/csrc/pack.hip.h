// pack.hip.h — parameter pack / unpack / device init, global pooling, stream probe
// (part of the single translation unit mfas_hip.hip; see the header comment there and DESIGN.md)
#pragma once
// ------------------------------------------------------------------------------------------------
// Parameter import / export / device init (tile-major <-> reference row-major state_dict order)
// ------------------------------------------------------------------------------------------------
#define PK_SET 0
#define PK_GET 1
#define PK_INIT 2
#define PK_WT 3     // re-derive the transposed OUT / HEAD tiles (wt arena) from plane 0
#define PK_PUT 4    // write ONE plane (sel_plane) from flat, leaving the other planes alone (state transfer between two layouts of the
                    // same population: PK_SET plane 0, then PK_PUT the Adam moments); the wt arena follows plane 0 unless a.wt is null

struct PackArgs {
    const SegDesc* desc;
    const CandDev* cands;
    float* plane;
    int64_t plane_stride;
    float* wt;
    float* flat;            // SET: source, GET: destination (one candidate)
    const uint32_t* seeds;  // INIT: per candidate
    int32_t mode, sel_plane;
    Geo g;
};

__device__ __forceinline__ uint32_t d_param_seed(uint32_t seed, uint32_t slot) {
    return (uint32_t)(((unsigned long long)seed * 1000003ULL + (unsigned long long)slot * 7919ULL + 17ULL) & 0x7FFFFFFFULL);
}
__device__ __forceinline__ uint32_t d_hash_h0(uint32_t seed) { return lowbias32(seed * 0x9E3779B9U + 0x7F4A7C15U); }
__device__ __forceinline__ float d_hash_u01(uint32_t h0, uint32_t idx) {
    return (float)(lowbias32(idx ^ h0) >> 8) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(256) k_pack(const PackArgs a) {
    const SegDesc d = a.desc[blockIdx.x];
    const int nkb = d.cc >> 4, nrb = d.rows_p >> 4;
    float* Wp = a.plane + d.w_off;
    uint32_t h0 = 0;
    if (a.mode == PK_INIT) h0 = d_hash_h0(d_param_seed(a.seeds[d.cand], d.init_seed));
    if (a.mode == PK_WT && d.wt_off < 0) return;
    for (int e = threadIdx.x; e < d.rows_p * d.cc; e += 256) {
        const int tile = e >> 8, within = e & 255, lane = within >> 2, q = within & 3;
        const int rb = tile / nkb, kb = tile - rb * nkb;
        const int r = (d.rb0 + rb) * 16 + (lane & 15);   // row inside the segment (row-split units start at row block rb0)
        const int k = d.k0 + kb * 16 + 4 * (lane >> 4) + q;   // column inside the segment
        const bool ok = r < d.rows && k < d.cols;
        const int64_t fidx = (int64_t)r * d.src_ld + d.src_col0 + k;
        if (a.mode == PK_GET) {
            if (ok) a.flat[d.src_off + fidx] = Wp[a.sel_plane * a.plane_stride + e];
            continue;
        }
        float val = 0.f;
        if (a.mode == PK_WT) {
            val = Wp[e];
        } else if (a.mode == PK_PUT) {
            if (ok) val = a.flat[d.src_off + fidx];
            Wp[a.sel_plane * a.plane_stride + e] = val;
            if (a.sel_plane != 0 || !a.wt) continue;
        } else {
            if (ok) {
                if (a.mode == PK_SET) val = a.flat[d.src_off + fidx];
                else val = (d_hash_u01(h0, (uint32_t)fidx) * 2.0f - 1.0f) * d.init_bound;
            }
            Wp[e] = val;
            Wp[a.plane_stride + e] = 0.f;
            Wp[2 * a.plane_stride + e] = 0.f;
        }
        if (d.wt_off >= 0) {
            const int l15 = lane & 15, lg = lane >> 4;
            float* T = a.wt + d.wt_off + ((int64_t)((d.k0 >> 4) + kb) * d.seg_nrb + d.rb0 + rb) * 256;
            T[((((l15 >> 2) * 16 + 4 * lg) + q) << 2) + (l15 & 3)] = val;
        }
    }
}

// vector parameters of one candidate (SET/GET) or of all candidates (INIT: blockIdx.x = candidate)
__global__ void __launch_bounds__(256) k_vec(const PackArgs a, int cand_fixed) {
    const int cand = cand_fixed >= 0 ? cand_fixed : blockIdx.x;
    const CandDev& cd = a.cands[cand];
    const Geo& g = a.g;
    float* P0 = a.plane + cd.vec_off;
    const int tid = threadIdx.x;
    const int nvec = MFAS_MAX_CELLS * g.vec_cell_stride + g.Cp;
    if (a.mode != PK_GET && a.mode != PK_PUT)
        for (int e = tid; e < nvec; e += 256) {   // zero everything first (padding, Adam state)
            P0[e] = 0.f;
            P0[a.plane_stride + e] = 0.f;
            P0[2 * a.plane_stride + e] = 0.f;
        }
    __syncthreads();
    float* P = P0 + ((a.mode == PK_GET || a.mode == PK_PUT) ? a.sel_plane * a.plane_stride : 0);
    const uint32_t seed = a.mode == PK_INIT ? a.seeds[cand] : 0;
    for (int i = 0; i < cd.L; ++i) {
        float* vb = P + i * g.vec_cell_stride;
        const float bound = (float)(1.0 / sqrt((double)cd.K_in[i]));
        const uint32_t hb = d_hash_h0(d_param_seed(seed, 2 * i + 1));
        for (int r = tid; r < g.R; r += 256) {
            if (a.mode == PK_SET || a.mode == PK_PUT) {
                vb[VEC_B * g.Rp + r] = a.flat[cd.f_b[i] + r];
                if (g.bn) {
                    vb[VEC_G * g.Rp + r] = a.flat[cd.f_bn[i] + r];
                    vb[VEC_BE * g.Rp + r] = a.flat[cd.f_bn[i] + g.R + r];
                    if (a.mode == PK_SET || a.sel_plane == 0) {      // running stats exist only in plane 0
                        vb[VEC_RM * g.Rp + r] = a.flat[cd.f_bn[i] + 2 * g.R + r];
                        vb[VEC_RV * g.Rp + r] = a.flat[cd.f_bn[i] + 3 * g.R + r];
                    }
                }
            } else if (a.mode == PK_GET) {
                a.flat[cd.f_b[i] + r] = vb[VEC_B * g.Rp + r];
                if (g.bn) {
                    a.flat[cd.f_bn[i] + r] = vb[VEC_G * g.Rp + r];
                    a.flat[cd.f_bn[i] + g.R + r] = vb[VEC_BE * g.Rp + r];
                    // running stats exist only in plane 0
                    a.flat[cd.f_bn[i] + 2 * g.R + r] = a.sel_plane == 0 ? vb[VEC_RM * g.Rp + r] : 0.f;
                    a.flat[cd.f_bn[i] + 3 * g.R + r] = a.sel_plane == 0 ? vb[VEC_RV * g.Rp + r] : 0.f;
                }
            } else {
                vb[VEC_B * g.Rp + r] = (d_hash_u01(hb, (uint32_t)r) * 2.0f - 1.0f) * bound;
                if (g.bn) {
                    vb[VEC_G * g.Rp + r] = 1.0f;
                    vb[VEC_RV * g.Rp + r] = 1.0f;
                }
            }
        }
        if (tid == 0) {
            if (a.mode == PK_SET || a.mode == PK_PUT) vb[5 * g.Rp] = a.flat[cd.f_alpha + i];
            else if (a.mode == PK_GET) a.flat[cd.f_alpha + i] = vb[5 * g.Rp];
            else if (g.alphas) {
                const uint32_t ha = d_hash_h0(d_param_seed(seed, 40 + i));
                const float u0 = d_hash_u01(ha, 0), u1 = d_hash_u01(ha, 1), u2 = d_hash_u01(ha, 2), u3 = d_hash_u01(ha, 3);
                vb[5 * g.Rp] = ((((u0 + u1) + (u2 + u3)) - 2.0f) * 1.7320508075688772f) * 0.1f;
            }
        }
    }
    {
        float* hb_ = P + g.vec_head;
        const float bound = (float)(1.0 / sqrt((double)g.R));
        const uint32_t hh = d_hash_h0(d_param_seed(seed, 11));
        for (int c = tid; c < g.C; c += 256) {
            if (a.mode == PK_SET || a.mode == PK_PUT) hb_[c] = a.flat[cd.f_bc + c];
            else if (a.mode == PK_GET) a.flat[cd.f_bc + c] = hb_[c];
            else hb_[c] = (d_hash_u01(hh, (uint32_t)c) * 2.0f - 1.0f) * bound;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_mt_uniform — torch's CPU random stream on the device (round 4).  The reference builds every candidate from torch's global
// generator (ntu_searchable.py:44: searchable_type(args, conf) -> nn.Linear.reset_parameters: kaiming_uniform_(a = sqrt 5) on the
// weight, U(+-1/sqrt(fan_in)) on the bias, both Tensor.uniform_(lo, hi)); train_sampled_models' default initialisation reproduces
// those draws per candidate under torch.manual_seed(seed_base + 2 + i).  On the host that is 1 M serial Mersenne-Twister draws per
// R = 128 candidate (0.7-1 ms; 3 % of the headline call, 20 % of a 50-candidate search call at R = 16).  Here one workgroup per
// candidate runs at::mt19937 itself — seeding (init_with_uint32), the 624-word twist in three dependent phases through LDS, the
// tempering — and at::uniform_real_distribution<float>: x = (y & (2^24 - 1)) * 2^-24, then x * (hi - lo) + lo, which torch's
// AVX2 / AVX512 CPU kernels evaluate as ONE fused multiply-add (verified against torch on the host at first use,
// mfas_amd/ntu_searchable.py; a mismatch falls back to the host path).  The values land in the candidate's FLAT parameter vector
// (reference state_dict order), from which k_pack / k_vec lay them out like any set_params call; the next 64 raw outputs of each
// stream go back to the host, which draws the alpha parameters from them (normal_distribution<double>, libm — mfas_hip.hip).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fill(float* p, float v, int64_t n) {
    for (int64_t i = threadIdx.x; i < n; i += 256) p[i] = v;
}

#define MT_MAX_SEG 12
struct MtCand {
    uint32_t seed;                      // torch.manual_seed(seed): at::mt19937 takes the low 32 bits
    int32_t nseg;
    int64_t total;                      // draws of this candidate
    int64_t start[MT_MAX_SEG + 1];      // first draw of segment j (prefix sums)
    int64_t dst[MT_MAX_SEG];            // flat-parameter offset of segment j's first element
    float lo[MT_MAX_SEG], hi[MT_MAX_SEG];
    int64_t flat_off;                   // this candidate's flat vector inside the scratch buffer
};
#define MT_TAIL 64

__global__ void __launch_bounds__(256) k_mt_uniform(const MtCand* cands, float* flat, uint32_t* tails) {
    __shared__ uint32_t st[2][624];
    const MtCand& c = cands[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid == 0) {                     // at::mt19937::init_with_uint32
        uint32_t x = c.seed;
        st[0][0] = x;
        for (int j = 1; j < 624; ++j) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)j; st[0][j] = x; }
    }
    __syncthreads();
    float* out = flat + c.flat_off;
    uint32_t* tail = tails + (size_t)blockIdx.x * MT_TAIL;
    int cur = 0, seg = 0;
    auto twist = [](uint32_t a, uint32_t b, uint32_t m) {
        const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    };
    for (int64_t pos = 0; pos < c.total + MT_TAIL; pos += 624) {
        const uint32_t* o = st[cur];
        uint32_t* n = st[cur ^ 1];
        if (tid < 227) n[tid] = twist(o[tid], o[tid + 1], o[tid + 397]);                 // i in [0, 227): old words only
        __syncthreads();
        if (tid < 227) n[227 + tid] = twist(o[227 + tid], o[228 + tid], n[tid]);        // i in [227, 454): new word i - 227
        __syncthreads();
        if (tid < 170) {                                                                 // i in [454, 624): new word i - 227; i = 623 wraps to the NEW word 0
            const int i = 454 + tid;
            n[i] = twist(o[i], i == 623 ? n[0] : o[i + 1], n[i - 227]);
        }
        __syncthreads();
        cur ^= 1;
        for (int j = tid; j < 624; j += 256) {
            uint32_t y = n[j];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            const int64_t d = pos + j;
            if (d < c.total) {
                int sj = seg;                                   // (segments are visited in order: resume from this block's first)
                while (d >= c.start[sj + 1]) ++sj;
                const float x = (float)(y & 0xFFFFFFu) * (1.0f / 16777216.0f);
                out[c.dst[sj] + (d - c.start[sj])] = __builtin_fmaf(x, c.hi[sj] - c.lo[sj], c.lo[sj]);
            } else if (d < c.total + MT_TAIL) {
                tail[d - c.total] = y;
            }
        }
        while (seg + 1 < c.nseg && pos + 624 >= c.start[seg + 1]) ++seg;   // the next block starts in (or after) this segment
        // (n is rewritten two blocks from now; the reads above are done before the next block's barriers release its writers)
    }
}

// ------------------------------------------------------------------------------------------------
// k_pool — GlobalPooling2D (models/auxiliary/aux_models.py:54-64): mean over all trailing dims of a (B, C, ...) tap.
// One wave per (b, c) row of `inner` contiguous elements, 16 B per lane per load, f32 accumulation, wave shuffle
// reduction; pure HBM-bound reduction (the "step before the path" that builds the feature table).
// ------------------------------------------------------------------------------------------------
template <typename T> struct PoolVec;
template <> struct PoolVec<float> { static constexpr int N = 4; };
template <> struct PoolVec<uint16_t> { static constexpr int N = 8; };

__device__ __forceinline__ float pool_cvt(uint16_t v, int dtype) {
    return dtype == MFAS_DT_BF16 ? __uint_as_float((uint32_t)v << 16) : __half2float(__ushort_as_half(v));
}

__global__ void __launch_bounds__(256) k_pool(const void* x, int dtype, int64_t rows, int64_t inner, void* out, int out_dtype) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float acc = 0.f;
    if (dtype == MFAS_DT_F32) {
        const float* p = reinterpret_cast<const float*>(x) + row * inner;
        const int64_t nv = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? inner / 4 : 0;
        for (int64_t i = lane; i < nv; i += 64) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
        for (int64_t i = nv * 4 + lane; i < inner; i += 64) acc += p[i];
    } else {
        const uint16_t* p = reinterpret_cast<const uint16_t*>(x) + row * inner;
        const int64_t nv = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? inner / 8 : 0;
        for (int64_t i = lane; i < nv; i += 64) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc += pool_cvt((uint16_t)(w[j] & 0xFFFFU), dtype) + pool_cvt((uint16_t)(w[j] >> 16), dtype);
        }
        for (int64_t i = nv * 8 + lane; i < inner; i += 64) acc += pool_cvt(p[i], dtype);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const float m = acc / (float)inner;
        if (out_dtype == MFAS_DT_F32) reinterpret_cast<float*>(out)[row] = m;
        else if (out_dtype == MFAS_DT_BF16) {
            uint32_t u = __float_as_uint(m);
            u += 0x7FFFU + ((u >> 16) & 1U);          // round to nearest even
            reinterpret_cast<uint16_t*>(out)[row] = (uint16_t)(u >> 16);
        } else reinterpret_cast<__half*>(out)[row] = __float2half(m);
    }
}

// ------------------------------------------------------------------------------------------------
// k_stream_probe — what this box's memory system gives the sweep's access pattern with NO compute: every wave
// read-modify-writes runs of 1 KiB tiles of three planes (16 B/lane, nontemporal), like W / m / v.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_stream_probe(float* P, size_t plane, size_t ntiles) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    constexpr int RUN = 8, U = 4;
    for (size_t r0 = wave * RUN; r0 < ntiles; r0 += nwaves * RUN)
        for (int t0 = 0; t0 < RUN && r0 + t0 < ntiles; t0 += U) {
            f32x4 w[U], m[U], v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t off = (r0 + t0 + u) * 256 + lane * 4;
                w[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + off));
                m[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + plane + off));
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(P + 2 * plane + off));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t off = (r0 + t0 + u) * 256 + lane * 4;
                __builtin_nontemporal_store(w[u] * 0.999f + m[u] * 0.001f, reinterpret_cast<f32x4*>(P + off));
                __builtin_nontemporal_store(m[u] * 0.9f + v[u], reinterpret_cast<f32x4*>(P + plane + off));
                __builtin_nontemporal_store(v[u] * 0.999f + w[u], reinterpret_cast<f32x4*>(P + 2 * plane + off));
            }
        }
}
