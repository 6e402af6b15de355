/*
 * plk_mfma.h -- K2+K3 for larger state spaces (9 <= k <= 64: amino acids, codons) on the
 * fp64 matrix cores.  Included by plk_engine.hip.
 *
 * For k = 20 or 61 the per-edge update L <- P_e L is a (k x k) x (k x sites) product with
 * arithmetic intensity above the fp64 ridge, so it is compute bound and GEMM shaped: it
 * goes on v_mfma_f64_16x16x4_f64.  Layout (MI355X guide, "f64 MFMA"):
 *     A: lane l holds A[row l&15][k l>>4]      B: lane l holds B[k l>>4][col l&15]
 *     D: lane l holds D[row (l>>4) + 4 r][col l&15] in result register r = 0..3
 * A wavefront owns 16 sites (columns); a site's partial vector is spread over the 4 lanes
 * {s, s+16, s+32, s+48}: lane group g = l>>4 holds states g, g+4, g+8, ... in registers.
 * With that layout the D tile t, register r of one product IS the B operand of k-step
 * 4t + r of the next product: chained matrix-vector products need no lane movement.
 * P_e is pre-arranged in "A fragments" (64 doubles per (row tile, k-step), lane order)
 * and staged through LDS once per op for the whole workgroup.
 *
 * The traversal program is the same post-order program as the other ll kernels
 * (OP_* in plk_engine.hip); stack slots live in HBM in the same distributed layout;
 * leaves use tip tables P_e * defs[code] (built in double-double) gathered from L2.
 *
 * Replaces: src/arbplfll.c:139-170 x src/evaluate_site_lhood.c:21-57 x src/util.c:242-301.
 */
#ifndef PLK_MFMA_H
#define PLK_MFMA_H

typedef double plk_d4 __attribute__((ext_vector_type(4)));

#ifndef MF_BLOCK
#define MF_BLOCK 256           /* 4 waves = 64 sites per workgroup (8 waves measured slower: profiles/r02_exp_codon_kernel_variants.json) */
#endif
#define MF_SITES (MF_BLOCK / 4)

struct MfmaArgs {
    long S, Spad;
    int k, kk4, C, nops, ntips, nchar, root_mode;
    const int4 *ops;          /* x = opcode | tip<<8, y = staged code row (observation ops) / slot */
    const int *obs_nodes;     /* nobs nodes whose code rows are staged in LDS */
    int nobs, first_slot, first_row;   /* first observation op of the program (first_slot < 0: none) */
    int first_mv;             /* op index of the first MATVEC (-1: none): its fragments are requested before the loop */
    const double *frag;       /* [C][nops][T][kk4][64] A fragments of P (zero padded) */
    const double *tip;        /* [C][ntips+1][nchar][4][4T]: [lane group][register], last slot = raw definitions */
    const uint8_t *codes;     /* [N][Spad] */
    const double *cat_prior;
    const double *root_wd;    /* [4][4T] root weights in the distributed layout */
    const double *w;
    double *slots;            /* [nslots][4T][Spad4] with Spad4 = 4 * padded sites (lane-linear) */
    long slot_stride;         /* doubles per (slot, register) plane */
    double *site_ll;
    dd *partial;
};

/* pattern codes of the block's 64 sites for every observed node: 16 dwords per row; eight (node lookup, code dword)
 * pairs per lane are in flight at a time instead of one */
__device__ static inline void mf_stage_codes(uint8_t *code_lds, const uint8_t *codes, const int *obs_nodes, int nobs, long Spad,
                                             size_t site0, int tid, int sites = MF_SITES)
{
    uint32_t *dst = reinterpret_cast<uint32_t *>(code_lds);
    const int ndw = nobs * (sites / 4);
    for (int i0 = tid; i0 < ndw; i0 += 8 * MF_BLOCK) {
        int node[8];
        uint32_t q[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int idx = i0 + u * MF_BLOCK; node[u] = idx < ndw ? obs_nodes[idx / (sites / 4)] : 0; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int idx = i0 + u * MF_BLOCK, col = idx % (sites / 4);
            q[u] = idx < ndw ? reinterpret_cast<const uint32_t *>(codes + (size_t)node[u] * Spad + site0)[col] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int idx = i0 + u * MF_BLOCK; if (idx < ndw) dst[idx] = q[u]; }
    }
}

/* A fragments of one matrix, L2 -> LDS, by the whole workgroup.  Eight 16-byte loads per lane are in flight before the
 * first one is stored: as a plain copy loop the compiler emitted load, wait, store per iteration -- eight serialised L2
 * round trips per staged matrix at k = 61, longer than the 64 matrix-core instructions that consume it. */
__device__ static inline void mf_stage(double *lds_frag, const double *src, int nfrag, int tid)
{
    __syncthreads();
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    double2 *d2 = reinterpret_cast<double2 *>(lds_frag);
    const int n2 = nfrag / 2;
    for (int i0 = tid; i0 < n2; i0 += 8 * MF_BLOCK) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = i0 + u * MF_BLOCK < n2 ? s2[i0 + u * MF_BLOCK] : double2{0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 8; u++) if (i0 + u * MF_BLOCK < n2) d2[i0 + u * MF_BLOCK] = v[u];
    }
    __syncthreads();
}

/* The same in two halves around the previous product: mf_request() issues this lane's eight 16-byte loads of the NEXT
 * matrix into registers and returns at once; mf_commit() -- after the product that still reads the current fragments --
 * writes them to LDS between two barriers.  The L2 round trip of a staged matrix (longer than the 64 matrix-core
 * instructions that consume it at k = 61) is then covered by the product before it, at the price of 32 VGPRs. */
struct MfPending { double2 v[8]; };
__device__ __forceinline__ void mf_request(MfPending &p, const double *src, int nfrag, int tid)
{
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    const int n2 = nfrag / 2;
#pragma unroll
    for (int u = 0; u < 8; u++) p.v[u] = tid + u * MF_BLOCK < n2 ? s2[tid + u * MF_BLOCK] : double2{0.0, 0.0};
}
__device__ __forceinline__ void mf_commit(double *lds_frag, const MfPending &p, int nfrag, int tid)
{
    double2 *d2 = reinterpret_cast<double2 *>(lds_frag);
    const int n2 = nfrag / 2;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; u++) if (tid + u * MF_BLOCK < n2) d2[tid + u * MF_BLOCK] = p.v[u];
    __syncthreads();
}

template <int T>
__device__ __forceinline__ void ll_mfma_body(const MfmaArgs &a)
{
    extern __shared__ double lds_frag[];          /* T * kk4 * 64 doubles, then nobs x 64 staged pattern codes */
    constexpr int R = 4 * T;                      /* registers (states) per lane */
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;                      /* lane group = state residue mod 4 */
    const long site = (long)blockIdx.x * MF_SITES + wave * 16 + (lane & 15);
    const bool valid = site < a.S;
    const long lin = ((long)blockIdx.x * MF_SITES + wave * 16) * 4 + lane;   /* lane-linear index for slots */
    const int nfrag = T * a.kk4 * 64;

    /* stage the code rows of this block's 64 sites (rows are padded to Spad, a multiple of 1024) */
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_frag + nfrag);
    mf_stage_codes(code_lds, a.codes, a.obs_nodes, a.nobs, a.Spad, (size_t)blockIdx.x * MF_SITES, tid);
    __syncthreads();
    const int scol = wave * 16 + (lane & 15);

    double sum = 0.0;
    int Eexp = 0;
    bool have = false;
    MfPending pend;
    mf_request(pend, a.frag + (size_t)(a.first_mv >= 0 ? a.first_mv : 0) * nfrag, nfrag, tid);

    for (int c = 0; c < a.C; c++) {
        double x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = 1.0;
        int esc = 0;
        for (int pc = 0; pc < a.nops; pc++) {
            int4 op;
            op.x = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc];
            op.y = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc + 1];
            op.z = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc + 2];
            op.w = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc + 3];
            const int code = op.x & 0xff;
            if (code == OP_MATVEC) {
                /* the A fragments of this edge (same for all 4 waves) were requested before the previous product; put
                 * them into LDS and request those of the next product (the next category's first one at the end) */
                mf_commit(lds_frag, pend, nfrag, tid);
                const int nc = op.z <= pc ? c + 1 : c;
                mf_request(pend, a.frag + ((size_t)(nc < a.C ? nc : c) * a.nops + op.z) * nfrag, nfrag, tid);
                plk_d4 acc[T];
#pragma unroll
                for (int t = 0; t < T; t++) acc[t] = (plk_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < R; q++) {
                    if (q < a.kk4) {
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            const double af = lds_frag[(t * a.kk4 + q) * 64 + lane];
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, x[q], acc[t], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < T; t++) {
                    x[4 * t + 0] = acc[t][0]; x[4 * t + 1] = acc[t][1];
                    x[4 * t + 2] = acc[t][2]; x[4 * t + 3] = acc[t][3];
                }
            } else if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
                const int t = code == OP_NODE_MUL ? a.ntips : (op.x >> 8);
                const int ch = code_lds[op.y * MF_SITES + scol];
                const double2 *tp = reinterpret_cast<const double2 *>(
                    a.tip + ((((size_t)c * (a.ntips + 1) + t) * a.nchar + ch) * 4 + g) * R);
                if (code == OP_TIP_SET) {
#pragma unroll
                    for (int r = 0; r < R; r += 2) { const double2 v = tp[r >> 1]; x[r] = v.x; x[r + 1] = v.y; }
                } else {
#pragma unroll
                    for (int r = 0; r < R; r += 2) { const double2 v = tp[r >> 1]; x[r] *= v.x; x[r + 1] *= v.y; }
                }
            } else if (code == OP_PUSH) {
                double *sp = a.slots + (size_t)op.y * R * a.slot_stride + lin;
#pragma unroll
                for (int r = 0; r < R; r++) sp[(size_t)r * a.slot_stride] = x[r];
            } else if (code == OP_POPMUL) {
                const double *sp = a.slots + (size_t)op.y * R * a.slot_stride + lin;
#pragma unroll
                for (int r = 0; r < R; r++) x[r] *= sp[(size_t)r * a.slot_stride];
            } else if (code == OP_SCALE) {
                double m = 0.0;
#pragma unroll
                for (int r = 0; r < R; r++) m = fmax(m, x[r]);
                m = fmax(m, __shfl_xor(m, 16, 64));
                m = fmax(m, __shfl_xor(m, 32, 64));
                const int e = frexp_exp(m);
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = ldexp(x[r], -e);
                esc += e;
            }
        }
        /* root expectation: weights in the same distributed layout, then across the 4 lanes of the site */
        double lh = 0.0;
        const double *rw = a.root_wd + g * R;
#pragma unroll
        for (int r = 0; r < R; r++) lh = fma(rw[r], x[r], lh);
        lh += __shfl_xor(lh, 16, 64);
        lh += __shfl_xor(lh, 32, 64);
        const double term = as_uniform(a.cat_prior)[c] * lh;
        if (term != 0.0) {
            if (!have) { sum = term; Eexp = esc; have = true; }
            else if (esc > Eexp) { sum = ldexp(sum, Eexp - esc) + term; Eexp = esc; }
            else sum += ldexp(term, esc - Eexp);
        }
    }
    const double ll = have ? log(sum) + (double)Eexp * 0.6931471805599453094 : -INFINITY;
    dd v = dd_make(0.0, 0.0);
    if (valid && g == 0) {
        if (a.site_ll) a.site_ll[site] = ll;
        v = a.w ? dd_two_prod(a.w[site], ll) : dd_make(ll, 0.0);
    }
    if (a.partial) {
        dd r = dd_block_sum(v);
        if (tid == 0) a.partial[blockIdx.x] = r;
    }
}

/*
 * The same kernel with TWO groups of 16 sites per wave (round 3; opt-in, PLK_OPT_MFMA_NS2 = 1: measured equal to the
 * one-group kernel at BASELINE config 5, 5.79 against 5.62 ms -- profiles/r03_exp_codon_kernel_variants.json): every A
 * fragment read from LDS feeds two matrix-core
 * instructions, a staged matrix (two workgroup barriers, 30 KB from L2 at k = 61) serves 128 sites instead of 64, and
 * the two groups' accumulation chains alternate in the matrix pipe.  Site of group j: block * 128 + 64 j + 16 wave + lane % 16;
 * staged code rows are 128 bytes; the stack slots keep the lane-linear layout (group j at + 256 j).
 */
template <int T>
__device__ __forceinline__ void ll_mfma_body2(const MfmaArgs &a)
{
    extern __shared__ double lds_frag[];          /* T * kk4 * 64 doubles, then nobs x 128 staged pattern codes */
    constexpr int R = 4 * T, NS = 2, SITES = NS * MF_SITES;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    const int nfrag = T * a.kk4 * 64;
    uint8_t *code_lds = reinterpret_cast<uint8_t *>(lds_frag + nfrag);
    mf_stage_codes(code_lds, a.codes, a.obs_nodes, a.nobs, a.Spad, (size_t)blockIdx.x * SITES, tid, SITES);
    __syncthreads();
    const int scol = wave * 16 + (lane & 15);
    long site[NS], lin[NS];
#pragma unroll
    for (int j = 0; j < NS; j++) {
        site[j] = (long)blockIdx.x * SITES + j * MF_SITES + scol;
        lin[j] = ((long)blockIdx.x * SITES + j * MF_SITES + wave * 16) * 4 + lane;
    }
    double sum[NS] = {0.0, 0.0};
    int Eexp[NS] = {0, 0};
    bool have[NS] = {false, false};
    MfPending pend;
    mf_request(pend, a.frag + (size_t)(a.first_mv >= 0 ? a.first_mv : 0) * nfrag, nfrag, tid);

    for (int c = 0; c < a.C; c++) {
        double x[NS][R];
#pragma unroll
        for (int j = 0; j < NS; j++)
#pragma unroll
            for (int r = 0; r < R; r++) x[j][r] = 1.0;
        int esc[NS] = {0, 0};
        for (int pc = 0; pc < a.nops; pc++) {
            int4 op;
            op.x = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc];
            op.y = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc + 1];
            op.z = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc + 2];
            op.w = as_uniform(reinterpret_cast<const int *>(a.ops))[4 * pc + 3];
            const int code = op.x & 0xff;
            if (code == OP_MATVEC) {
                mf_commit(lds_frag, pend, nfrag, tid);
                const int nc = op.z <= pc ? c + 1 : c;
                mf_request(pend, a.frag + ((size_t)(nc < a.C ? nc : c) * a.nops + op.z) * nfrag, nfrag, tid);
                plk_d4 acc[NS][T];
#pragma unroll
                for (int j = 0; j < NS; j++)
#pragma unroll
                    for (int t = 0; t < T; t++) acc[j][t] = (plk_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < R; q++) {
                    if (q < a.kk4) {
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            const double af = lds_frag[(t * a.kk4 + q) * 64 + lane];
#pragma unroll
                            for (int j = 0; j < NS; j++) acc[j][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, x[j][q], acc[j][t], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < NS; j++)
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        x[j][4 * t + 0] = acc[j][t][0]; x[j][4 * t + 1] = acc[j][t][1];
                        x[j][4 * t + 2] = acc[j][t][2]; x[j][4 * t + 3] = acc[j][t][3];
                    }
            } else if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
                const int t = code == OP_NODE_MUL ? a.ntips : (op.x >> 8);
#pragma unroll
                for (int j = 0; j < NS; j++) {
                    const int ch = code_lds[op.y * SITES + j * MF_SITES + scol];
                    const double2 *tp = reinterpret_cast<const double2 *>(
                        a.tip + ((((size_t)c * (a.ntips + 1) + t) * a.nchar + ch) * 4 + g) * R);
                    if (code == OP_TIP_SET) {
#pragma unroll
                        for (int r = 0; r < R; r += 2) { const double2 v = tp[r >> 1]; x[j][r] = v.x; x[j][r + 1] = v.y; }
                    } else {
#pragma unroll
                        for (int r = 0; r < R; r += 2) { const double2 v = tp[r >> 1]; x[j][r] *= v.x; x[j][r + 1] *= v.y; }
                    }
                }
            } else if (code == OP_PUSH) {
#pragma unroll
                for (int j = 0; j < NS; j++) {
                    double *sp = a.slots + (size_t)op.y * R * a.slot_stride + lin[j];
#pragma unroll
                    for (int r = 0; r < R; r++) sp[(size_t)r * a.slot_stride] = x[j][r];
                }
            } else if (code == OP_POPMUL) {
#pragma unroll
                for (int j = 0; j < NS; j++) {
                    const double *sp = a.slots + (size_t)op.y * R * a.slot_stride + lin[j];
#pragma unroll
                    for (int r = 0; r < R; r++) x[j][r] *= sp[(size_t)r * a.slot_stride];
                }
            } else if (code == OP_SCALE) {
#pragma unroll
                for (int j = 0; j < NS; j++) {
                    double m = 0.0;
#pragma unroll
                    for (int r = 0; r < R; r++) m = fmax(m, x[j][r]);
                    m = fmax(m, __shfl_xor(m, 16, 64));
                    m = fmax(m, __shfl_xor(m, 32, 64));
                    const int e = frexp_exp(m);
#pragma unroll
                    for (int r = 0; r < R; r++) x[j][r] = ldexp(x[j][r], -e);
                    esc[j] += e;
                }
            }
        }
        const double *rw = a.root_wd + g * R;
#pragma unroll
        for (int j = 0; j < NS; j++) {
            double lh = 0.0;
#pragma unroll
            for (int r = 0; r < R; r++) lh = fma(rw[r], x[j][r], lh);
            lh += __shfl_xor(lh, 16, 64);
            lh += __shfl_xor(lh, 32, 64);
            const double term = as_uniform(a.cat_prior)[c] * lh;
            if (term != 0.0) {
                if (!have[j]) { sum[j] = term; Eexp[j] = esc[j]; have[j] = true; }
                else if (esc[j] > Eexp[j]) { sum[j] = ldexp(sum[j], Eexp[j] - esc[j]) + term; Eexp[j] = esc[j]; }
                else sum[j] += ldexp(term, esc[j] - Eexp[j]);
            }
        }
    }
    dd v = dd_make(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < NS; j++) {
        const double ll = have[j] ? log(sum[j]) + (double)Eexp[j] * 0.6931471805599453094 : -INFINITY;
        if (site[j] < a.S && g == 0) {
            if (a.site_ll) a.site_ll[site[j]] = ll;
            v = dd_add(v, a.w ? dd_two_prod(a.w[site[j]], ll) : dd_make(ll, 0.0));
        }
    }
    if (a.partial) {
        dd r = dd_block_sum(v);
        if (tid == 0) a.partial[blockIdx.x] = r;
    }
}

template <int T>
__global__ __launch_bounds__(MF_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ll_mfma_ns2(MfmaArgs a) { ll_mfma_body2<T>(a); }

template <int T>
__global__ __launch_bounds__(MF_BLOCK) void k_ll_mfma(MfmaArgs a) { ll_mfma_body<T>(a); }

/* the same kernel compiled for 4 waves per SIMD (<= 128 registers): +25 % at k = 61 (T = 4), where the default
 * allocation of 160 registers leaves 3 waves; slower at T = 2, which already fits */
template <int T>
__global__ __launch_bounds__(MF_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ll_mfma_occ4(MfmaArgs a) { ll_mfma_body<T>(a); }

/* A fragments: frag[((c*nops + pc)*T + t)*kk4 + q][l] = P[c][edge(pc)][16t + (l&15)][4q + (l>>4)] */
__global__ void k_build_frag(int k, int T, int kk4, int E, int nops, const int *__restrict__ op_edge,
                             const double *__restrict__ P, double *__restrict__ frag)
{
    const int pc = blockIdx.x, c = blockIdx.y;
    const int e = op_edge[pc];
    double *dst = frag + ((size_t)c * nops + pc) * T * kk4 * 64;
    const int n = T * kk4 * 64;
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const int l = idx & 63, tq = idx >> 6;
        const int t = tq / kk4, q = tq - t * kk4;
        const int i = 16 * t + (l & 15), j = 4 * q + (l >> 4);
        double v = 0.0;
        if (e >= 0 && i < k && j < k) v = P[((size_t)c * E + e) * k * k + (size_t)i * k + j];
        dst[idx] = v;
    }
}

/* distributed tip tables: tip[(((c*(ntips+1) + t)*nchar + code)*4 + g)*R + r] = (P_e defs[code])[g + 4r]
 * (dd accumulation; exact for constant definition rows; slot ntips holds defs[code] itself).  A definition row that is
 * zero except for one 1.0 (an observed state: all but a few rows of a real alphabet) selects a column of P_e: the
 * double-double sum then has one non-zero term and returns it unchanged, so the column is copied instead of summed
 * (the same bits, 1/k of the work: 262 -> see profiles/r03_exp_codon_kernel_variants.json). */
#define PLK_DEF_KIND_CACHE 1024
__device__ static inline int def_row_kind(const double *d, int k)
{
    bool constant = true;
    int nnz = 0, at = -1;
    for (int j = 0; j < k; j++) {
        constant = constant && (d[j] == d[0]);
        if (d[j] != 0.0) { nnz++; at = j; }
    }
    if (constant) return -2;                    /* src/util.c:276-283: a constant row maps to itself */
    return nnz == 1 && d[at] == 1.0 ? at : -1;
}

__global__ void k_build_tip_dist(int k, int R, int E, int ntips, int nchar, const int *__restrict__ tip_edge,
                                 const dd *__restrict__ Pdd, const double *__restrict__ defs /* [nchar][K] */, int Kpad,
                                 double *__restrict__ tip)
{
    __shared__ int s_kind[PLK_DEF_KIND_CACHE];
    const int t = blockIdx.x, c = blockIdx.y;
    const int e = tip_edge[t];
    const int n = nchar * 4 * R;
    for (int code = threadIdx.x; code < nchar && code < PLK_DEF_KIND_CACHE; code += blockDim.x) s_kind[code] = def_row_kind(defs + (size_t)code * Kpad, k);
    __syncthreads();
    for (int idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const int r = idx % R, gq = idx / R;
        const int g = gq & 3, code = gq >> 2;
        const int i = g + 4 * r;
        const double *d = defs + (size_t)code * Kpad;
        double out = 0.0;
        if (i < k) {
            if (e < 0) out = d[i];
            else {
                const int kind = code < PLK_DEF_KIND_CACHE ? s_kind[code] : def_row_kind(d, k);
                const dd *Pm = Pdd + ((size_t)c * E + e) * k * k + (size_t)i * k;
                if (kind == -2) out = d[0];
                else if (kind >= 0) out = Pm[kind].hi;
                else {
                    dd acc = dd_make(0.0, 0.0);
                    for (int j = 0; j < k; j++) acc = dd_add(acc, dd_mul_d(Pm[j], d[j]));
                    out = acc.hi;
                }
            }
        }
        tip[(((size_t)c * (ntips + 1) + t) * nchar + code) * 4 * R + (size_t)g * R + r] = out;
    }
}

#endif
