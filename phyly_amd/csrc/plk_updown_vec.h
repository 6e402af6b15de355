/*
 * plk_updown_vec.h -- down pass with stored vectors and BFS-order up pass for medium state spaces
 * (9 <= k <= 20: amino acids) on the vector fp64 pipe: arbplf-deriv / -marginal / -dwell / -trans / -em-update
 * at K = 16 and K = 20.  Included by plk_engine.hip.
 *
 * Same formulas as k_down_store / k_up (src/evaluate_site_lhood.c:7-63 with node vectors kept,
 * src/evaluate_site_forward.c:32-105, src/arbplfderiv.c:112-207,:312-342, src/evaluate_site_marginal.c:7-21,
 * src/arbplfmarginal.c:206-234, src/evaluate_site_frechet.c:5-42).  Round 1 ran these state counts on the
 * matrix cores in the 4-lanes-per-site layout: k = 20 was padded to two 16-row tiles (2.56x the matrix work) and
 * every stored vector to 32 doubles (1.6x the traffic), 0.28 of the HBM roof at BASELINE config 4.  Here:
 *   - one site per lane, a vector in K VGPR pairs, P_e / M_e / P_e^T as scalar-loaded SGPR operands of v_fma_f64
 *     (k^2 FMAs per product, no padding, no LDS, no barriers) -- the k_ll_vec design;
 *   - stored vectors un-padded, one plane per state: [entity][category][state][site], 512 contiguous bytes per
 *     wavefront access; only internal node vectors L_a and forward vectors F_a are stored, edge vectors
 *     P_e L_b are recomputed in the up pass from the L_b the edge form needs anyway;
 *   - down pass = depth-first traversal of the post-order program: every L_a is written once, at the product that
 *     consumes it; vectors waiting for a sibling subtree go to HBM stack slots (they come back from L2 /
 *     Infinity Cache);
 *   - up pass: one edge at a time with ~130 live registers (3-4 waves per SIMD); leaf edges gather P_e B_b and
 *     M_e B_b from double-double built tables;
 *   - the matrices of the up pass are laid out as a stream in the exact order of use (built per query), so the
 *     64-byte lines of the next product's matrix are requested through the scalar cache while the current product
 *     runs (vec_touch of plk_vec.h: waited for inside the asm statement);
 *   - exact power-of-two rescaling with stored factors, categories combined at a common exponent (as k_up4).
 */
#ifndef PLK_UPDOWN_VEC_H
#define PLK_UPDOWN_VEC_H

#define UDV_BLOCK 128

struct UpVecArgs {
    long S, Spad, s0, n;
    int N, E, k, C, nchar, ntips, root_mode;
    int dzero;                     /* 1: edge-form matrices have zero row sums (dP) */
    /* down pass: program (int4: observation y = node; MATVEC y = CSR edge, z = storage index of the child node,
     * w = CSR edge of the next MATVEC (wrapping); PUSH / POPMUL y = slot; SCALE y = rescaling slot or -1);
     * plk_chain_build mode 3: observation ops y = staged row of the observation after next, z = tip slot of the next one
     * (bit 30: next category), w = its row */
    const int4 *ops;
    int nops, root_int;
    int first_slot, first_row, second_row;
    const double *PT;              /* [C][E][K*K] transposed P: PT[j*K+i] = P[i][j] (down pass) */
    const double *tip, *dtip;      /* [C][ntips+1][nchar][K]: P_e defs (slot ntips: defs themselves); M_e defs */
    const uint8_t *codes;
    const double *cat_prior, *root_w;
    /* up pass: visit records and the matrix stream in order of use */
    const int *visits;             /* plk_up_visits_build() of plk_program.h */
    int nvisits;
    const double *MS;              /* [C][nstream + 3][K*K] */
    int nstream;
    double *LN, *FN;               /* [(ent*C + c)][K][n] */
    double *slots;                 /* [slot][K][n] */
    int reg_lo;                    /* k_down_vec_rs: stack slots reg_lo .. reg_lo + 2 live in accumulation registers */
    double *SC, *CW, *XC;          /* as Up4Args */
    double *LH, *DV, *MV;          /* [n], [E][n], [N][k][n] */
    double *MVS;                   /* site-summed marginals only: [(node * k + state)][nwaves] per-wave weighted sums, MV unused */
    const double *wsite;           /* [n] site weights of the chunk or null */
    const int *code_state;         /* [nchar]: the state a character code observes (definition row = one 1.0, zeros elsewhere), or -1 */
};

/* child record of a visit: 4 ints */
#define UDV_WANT_D PLK_UP_WANT_D
#define UDV_WANT_F PLK_UP_WANT_F   /* forward vector of the child is needed (internal child, or its marginal) */
#define UDV_WANT_M PLK_UP_WANT_M
#define UDV_STORE_F PLK_UP_STORE_F /* internal child: F_b is stored */

/* plane i of a stored vector starts at base + i * n (wave-uniform: scalar address arithmetic); the lane adds its
 * site index as a 32-bit offset, so one VGPR addresses all K planes (n < 2^29 sites per chunk) */
template <int K>
__device__ __forceinline__ void udv_load(const double *base, size_t n, long slc, double (&out)[K])
{
    const unsigned off = (unsigned)slc;
    /* pin the (wave-uniform) base in an SGPR pair here: otherwise the compiler hoists K 64-bit lane addresses per
     * vector out of the loops and keeps them in VGPRs (40 registers for one K = 20 vector) */
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int i = 0; i < K; i++) out[i] = (base + (size_t)i * n)[off];
}
template <int K>
__device__ __forceinline__ void udv_store(double *base, size_t n, long slc, const double (&v)[K])
{
    const unsigned off = (unsigned)slc;
    asm volatile("" : "+s"(base));
#pragma unroll
    for (int i = 0; i < K; i++) (base + (size_t)i * n)[off] = v[i];
}
template <int K>
__device__ __forceinline__ void udv_gather(const double *row, double (&out)[K])
{
    const double2 *tp = reinterpret_cast<const double2 *>(row);
#pragma unroll
    for (int i = 0; i < K; i += 2) { const double2 v = tp[i >> 1]; out[i] = v.x; out[i + 1] = v.y; }
}
/* all k entries equal?  (the reference's exact constant-column test, src/arb_mat_extras.c:36-51.)  Compared bit
 * for bit with integer operations: K - 1 floating-point compares would each produce a lane mask in an SGPR pair, and
 * the compiler kept them all live (845 spilled SGPRs in k_down_vec<20>).  +0.0 and -0.0 count as different here; the
 * general product then runs, which is still correct. */
template <int K>
__device__ __forceinline__ bool udv_const(const double (&x)[K], int k)
{
    const unsigned lo0 = (unsigned)__double2loint(x[0]), hi0 = (unsigned)__double2hiint(x[0]);
    unsigned diff = 0;
#pragma unroll
    for (int i = 1; i < K; i++)
        if (i < k) diff |= ((unsigned)__double2loint(x[i]) ^ lo0) | ((unsigned)__double2hiint(x[i]) ^ hi0);
    return diff == 0;
}
/* ---------------------------------------------------------------------------------------------------------- */
template <int K, int NREG>
__device__ __forceinline__ void k_down_vec_body(const UpVecArgs &a, const int *__restrict__ obs_nodes)
{
    if constexpr (NREG > 0) {
        static_assert(2 * NREG * K <= 128, "register stack: accumulation registers a0..a127");
        asm volatile("" ::: PLK_CLOBBER_A0_31, PLK_CLOBBER_A32_63, PLK_CLOBBER_A64_127);     /* the kernel owns a0..a127 (see k_ll_vec_rs) */
    }
    const long sl = (long)blockIdx.x * UDV_BLOCK + threadIdx.x;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    const PLK_AS4 int *ops = as_uniform(reinterpret_cast<const int *>(a.ops));
    const PLK_AS4 int *obs = as_uniform(obs_nodes);
    const PLK_AS4 double *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * K;
    int xmax = INT_MIN;
    /* observation prefetch chain, two ops deep (see k_ll_vec) */
    double nv[K];
    int code_next = 0;
    if (a.first_slot >= 0) {
        const int ch0 = a.codes[(size_t)obs[a.first_row] * a.Spad + sg];
        udv_gather<K>(a.tip + ((size_t)a.first_slot * a.nchar + ch0) * K, nv);
        code_next = a.codes[(size_t)obs[a.second_row] * a.Spad + sg];
    } else {
#pragma unroll
        for (int i = 0; i < K; i++) nv[i] = 1.0;
    }
    for (int c = 0; c < a.C; c++) {
        double cur[K];
#pragma unroll
        for (int i = 0; i < K; i++) cur[i] = 1.0;
        int X = 0;
        const PLK_AS4 double *PTc = as_uniform(a.PT) + (size_t)c * a.E * K * K;
        const double *tipc = a.tip + (size_t)c * tabc;
        for (int pc = 0; pc < a.nops; pc++) {
            const int ox = ops[4 * pc], oy = ops[4 * pc + 1];
            const int code = ox & 0xff;
            if (code == OP_MATVEC) {
                const int oz = ops[4 * pc + 2], ow = ops[4 * pc + 3];
                /* the child's vector is final: store it, then multiply by its edge's P */
                if (valid && oz >= 0) udv_store<K>(a.LN + ((size_t)oz * a.C + c) * K * n, n, slc, cur);   /* oz < 0: rebuilt by the up pass */
                vec_touch<K>(PTc + (size_t)ow * K * K);            /* lines of the next product's matrix */
                const bool cst = udv_const<K>(cur, a.k);
                const double x0 = cur[0];
                double acc[K];
                vec_matvec<K>(PTc + (size_t)oy * K * K, cur, acc);
#pragma unroll
                for (int i = 0; i < K; i++) cur[i] = cst ? (i < a.k ? x0 : 0.0) : acc[i];   /* src/util.c:276-283 */
            } else if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
                const int oz = ops[4 * pc + 2];
                if (code == OP_TIP_SET) {
#pragma unroll
                    for (int i = 0; i < K; i++) cur[i] = nv[i];
                } else {
#pragma unroll
                    for (int i = 0; i < K; i++) cur[i] *= nv[i];
                }
                const int wrap = (oz >> 30) & 1;
                if (!wrap || c + 1 < a.C)
                    udv_gather<K>(tipc + (size_t)wrap * tabc + ((size_t)(oz & 0x3fffffff) * a.nchar + code_next) * K, nv);
                code_next = a.codes[(size_t)obs[oy] * a.Spad + sg];
            } else if (NREG > 0 && code == OP_PUSH && oy - a.reg_lo >= 0 && oy - a.reg_lo < NREG) {
                const int r = oy - a.reg_lo;
                if (r == 0) vec_acc_push<0, K>(cur, std::make_integer_sequence<int, K>());
                else if (NREG > 1 && r == 1) vec_acc_push<(NREG > 1 ? 1 : 0), K>(cur, std::make_integer_sequence<int, K>());
                else vec_acc_push<(NREG > 2 ? 2 : 0), K>(cur, std::make_integer_sequence<int, K>());
            } else if (NREG > 0 && code == OP_POPMUL && oy - a.reg_lo >= 0 && oy - a.reg_lo < NREG) {
                const int r = oy - a.reg_lo;
                if (r == 0) vec_acc_popmul<0, K>(cur, std::make_integer_sequence<int, K>());
                else if (NREG > 1 && r == 1) vec_acc_popmul<(NREG > 1 ? 1 : 0), K>(cur, std::make_integer_sequence<int, K>());
                else vec_acc_popmul<(NREG > 2 ? 2 : 0), K>(cur, std::make_integer_sequence<int, K>());
            } else if (code == OP_PUSH) {
                if (valid) udv_store<K>(a.slots + (size_t)oy * K * n, n, slc, cur);
            } else if (code == OP_POPMUL) {
                double v[K];
                udv_load<K>(a.slots + (size_t)oy * K * n, n, slc, v);
#pragma unroll
                for (int i = 0; i < K; i++) cur[i] *= v[i];
            } else if (code == OP_SCALE) {
                if (oy >= 0) {
                    double mx = 0.0;
#pragma unroll
                    for (int i = 0; i < K; i++) mx = fmax(mx, cur[i]);
                    double sc = 1.0;
                    if (mx > 0x1p-1000 && mx < 0x1p+1000) {
                        const int ex = ilogb(mx);
                        sc = ldexp(1.0, -ex);
#pragma unroll
                        for (int i = 0; i < K; i++) cur[i] *= sc;
                        X += ex;
                    }
                    if (valid) a.SC[((size_t)oy * a.C + c) * n + slc] = sc;
                }
            }
        }
        if (valid) udv_store<K>(a.LN + ((size_t)a.root_int * a.C + c) * K * n, n, slc, cur);
        double lh_c = 0.0;
#pragma unroll
        for (int i = 0; i < K; i++) lh_c = fma(rw[i], cur[i], lh_c);        /* root_w is zero padded; NONE -> ones, UNIFORM -> 1/k */
        lh_c *= prior[c];
        if (lh_c > 0.0 && X > xmax) xmax = X;
        if (valid) { a.XC[(size_t)c * n + slc] = (double)X; a.CW[(size_t)c * n + slc] = lh_c; }
    }
    if (xmax == INT_MIN) xmax = 0;
    if (valid) {
        double lh_total = 0.0;
        for (int c = 0; c < a.C; c++) {
            const double w = ldexp(1.0, (int)a.XC[(size_t)c * n + slc] - xmax);
            lh_total = fma(a.CW[(size_t)c * n + slc], w, lh_total);
            a.CW[(size_t)c * n + slc] = w;
        }
        a.LH[sl] = lh_total;
    }
}

template <int K>
__global__ __launch_bounds__(UDV_BLOCK) void k_down_vec(UpVecArgs a, const int *__restrict__ obs_nodes) { k_down_vec_body<K, 0>(a, obs_nodes); }

/* (A register-stack variant like k_ll_vec_rs was built for this pass in round 3 and dropped: with the prefetched tip row
 * (K register pairs) live next to cur and acc the compiler itself spills into accumulation registers, which the explicit
 * AGPR stack cannot share -- tools/isa_lint.py refuses the build -- and the ll kernel's variant gained only 4 %:
 * profiles/r03_exp_aa_kernel_variants.json.  k_down_vec_body keeps the NREG parameter for a two-sites-per-lane follow-up.) */

/*
 * Up pass.  visits: for every internal node in BFS order
 *   header (8 ints): node a, number of children, storage index of a, rescaling slot or -1, has_data, 0, 0, 0
 *   per child (4 ints): child node b, tip slot (>= 0) / -1 internal / <= -2 internal and handled inline, UDV_* flags,
 *   storage index of b (internal) or -1;
 *   children with a CSR edge index idx = first_edge + position: the header's int [5] holds first_edge.
 * Matrix stream MS (per category): the matrices in the order the visit code below consumes them:
 *   per child in order: for every internal sibling PT(sibling); internal child with WANT_D: DT(child); WANT_F: PN(child)
 * (the host builder plk_up_visits_build() and this kernel are the two halves of that contract; the stream always has one
 * spare matrix at the end for the look-ahead)
 */
/* Sums over the wave's 64 lanes of K per-lane values (the K states of a node's weighted marginal term), through a
 * wave-private LDS strip instead of K butterfly reductions (K x 18 data-parallel-primitive instructions: a third of the
 * marginal pass at K = 20).  Two halves of H = K / 2 states: every lane writes its H values as rows [state][lane] (row
 * stride 65: the column reads below spread over the banks), then lane (q = lane / 16, s = lane % 16 < H) adds the 16
 * values of state s that lanes 16 q .. 16 q + 15 wrote, in lane order, and two exchanges add the four quarters.
 * Returns, in lanes with q < 2 and s < H, the total of state q * H + s (udv_sum_state() names it); fixed order, no atomics. */
#define UDV_SUM_STRIP(K_) (((K_) / 2) * 65)
template <int K>
__device__ __forceinline__ double udv_state_sums(const double (&v)[K], double *strip, int lane)
{
    constexpr int H = K / 2;
    static_assert(K % 2 == 0 && H <= 16, "state sums: K");
    const int q = lane >> 4, s = lane & 15;
    double res[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                    /* the strip's previous reads are done (same wave: in order) */
#pragma unroll
        for (int i = 0; i < H; i++) strip[i * 65 + lane] = v[h * H + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double acc = 0.0;
        const double *col = strip + (s < H ? s : 0) * 65 + q * 16;
#pragma unroll
        for (int j = 0; j < 16; j++) acc += col[j];
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        res[h] = acc;
    }
    return q == 0 ? res[0] : res[1];
}
/* the state whose total udv_state_sums() left in this lane, or -1 */
template <int K>
__device__ __forceinline__ int udv_sum_state(int lane)
{
    const int q = lane >> 4, s = lane & 15;
    return q < 2 && s < K / 2 ? q * (K / 2) + s : -1;
}

template <int K, bool DERIV, bool MARG>
__global__ __launch_bounds__(UDV_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_up_vec(UpVecArgs a)
{
    const long sl = (long)blockIdx.x * UDV_BLOCK + threadIdx.x;
    const bool valid = sl < a.n;
    const long slc = valid ? sl : a.n - 1;
    const long sg = a.s0 + slc;
    const size_t n = (size_t)a.n;
    const PLK_AS4 int *vis = as_uniform(a.visits);
    const PLK_AS4 double *prior = as_uniform(a.cat_prior), *rw = as_uniform(a.root_w);
    const size_t tabc = (size_t)(a.ntips + 1) * a.nchar * K;
    const double inv = 1.0 / a.LH[slc];
    constexpr int KK = K * K;
    const size_t nwv = (size_t)gridDim.x * (UDV_BLOCK / 64), wv = (size_t)blockIdx.x * (UDV_BLOCK / 64) + (threadIdx.x >> 6);
    __shared__ double sum_strips[MARG ? (UDV_BLOCK / 64) * UDV_SUM_STRIP(K) : 1];      /* site-summed marginals: one strip per wave (udv_state_sums) */
    double *strip = sum_strips + (MARG ? (threadIdx.x >> 6) * UDV_SUM_STRIP(K) : 0);
    const int sum_st = udv_sum_state<K>(threadIdx.x & 63);

    {   /* root: forward vector = root prior weights; its marginal */
        const int root = vis[0], root_int = vis[2];
        double macc[K];
#pragma unroll
        for (int i = 0; i < K; i++) macc[i] = 0.0;
        for (int c = 0; c < a.C; c++) {
            double f[K];
#pragma unroll
            for (int i = 0; i < K; i++) f[i] = rw[i];
            if (valid) udv_store<K>(a.FN + ((size_t)root_int * a.C + c) * K * n, n, slc, f);
            if (MARG) {
                double l[K];
                udv_load<K>(a.LN + ((size_t)root_int * a.C + c) * K * n, n, slc, l);
                const double pc = prior[c] * a.CW[(size_t)c * n + slc];
#pragma unroll
                for (int i = 0; i < K; i++) macc[i] = fma(pc * f[i], l[i], macc[i]);
            }
        }
        if (MARG && vis[6]) {
            if (a.MVS) {
                const double ws = valid ? (a.wsite ? a.wsite[sl] : 1.0) * inv : 0.0;
#pragma unroll
                for (int i = 0; i < K; i++) {
                    const double t_ = wave64_sum_lane63(macc[i] * ws);
                    if (i < a.k && (threadIdx.x & 63) == 63) a.MVS[((size_t)root * a.k + i) * nwv + wv] = t_;
                }
            } else if (valid) {
#pragma unroll
                for (int i = 0; i < K; i++)
                    if (i < a.k) a.MV[((size_t)root * a.k + i) * n + sl] = macc[i] * inv;
            }
        }
    }

    for (int c = 0; c < a.C; c++) {
        const double *tipc = a.tip + (size_t)c * tabc;
        const double *dtipc = a.dtip + (size_t)c * tabc;
        /* every product requests the lines of the matrix (or pair) two slots ahead: each slot is touched once, one or
         * two products before its use; three spare slots at the end of the stream */
        const PLK_AS4 double *ms = as_uniform(a.MS) + (size_t)c * (a.nstream + 3) * KK;
        const double pc = prior[c] * a.CW[(size_t)c * n + slc];
        const bool first_cat = c == 0, last_cat = c == a.C - 1;
        int vp = 0;
        for (int v = 0; v < a.nvisits; v++) {
            const int nd = vis[vp], deg = vis[vp + 1], nd_int = vis[vp + 2], slot = vis[vp + 3], hd = vis[vp + 4], e0 = vis[vp + 5];
            const PLK_AS4 int *ch = vis + vp + 8;
            vp += 8 + 4 * deg;
            /* accumulate over categories in the output planes: first category writes, later ones add */
            /* L vector of internal child B_ (child record fields T_, BI_) into OUT: read from LN, or -- for a child that is
             * finished inside this visit (T_ <= -2: its record q lists its one or two leaves) -- rebuilt from the tip
             * tables in the order the down pass multiplied it (leaf rows, own observation, rescaling factor): the same
             * bits, and the down pass does not store it */
#define UDV_CHILD_L(B_, T_, BI_, OUT)                                                                     \
            do { if ((T_) > -2) udv_load<K>(a.LN + ((size_t)(BI_) * a.C + c) * K * n, n, slc, OUT);      \
                 else { const PLK_AS4 int *q_ = vis + (-2 - (T_));                                        \
                        udv_gather<K>(tipc + ((size_t)q_[6] * a.nchar + a.codes[(size_t)q_[5] * a.Spad + sg]) * K, OUT); \
                        if (q_[3] == 2) { double r_[K];                                                   \
                            udv_gather<K>(tipc + ((size_t)q_[9] * a.nchar + a.codes[(size_t)q_[8] * a.Spad + sg]) * K, r_); \
                            _Pragma("unroll") for (int i = 0; i < K; i++) OUT[i] *= r_[i]; }              \
                        if (q_[1]) { double r_[K];                                                        \
                            udv_gather<K>(tipc + ((size_t)a.ntips * a.nchar + a.codes[(size_t)(B_) * a.Spad + sg]) * K, r_); \
                            _Pragma("unroll") for (int i = 0; i < K; i++) OUT[i] *= r_[i]; }              \
                        if (q_[2] >= 0) { const double s_ = a.SC[((size_t)q_[2] * a.C + c) * n + slc];   \
                            _Pragma("unroll") for (int i = 0; i < K; i++) OUT[i] *= s_; } } } while (0)
#define UDV_OUT_D(EDGE, VAL)                                                                              \
            do { if (valid) { double *dp_ = a.DV + (size_t)(EDGE) * n + sl;                               \
                 const double t_ = (VAL); *dp_ = (first_cat ? t_ : *dp_ + t_) * (last_cat ? inv : 1.0); } } while (0)
#define UDV_OUT_M(NODE, FB, LB)                                                                           \
            do { if (a.MVS) {      /* site sums only: the wave's weighted sum of this category's term, accumulated per wave */ \
                     const double wi_ = valid ? (a.wsite ? a.wsite[sl] : 1.0) * inv * pc : 0.0;          \
                     double v_[K];                                                                        \
                     _Pragma("unroll") for (int i = 0; i < K; i++) v_[i] = wi_ * FB[i] * LB[i];           \
                     const double t_ = udv_state_sums<K>(v_, strip, threadIdx.x & 63);                   \
                     if (sum_st >= 0 && sum_st < a.k) {                                                   \
                         double *mp_ = a.MVS + ((size_t)(NODE) * a.k + sum_st) * nwv + wv;                \
                         *mp_ = first_cat ? t_ : *mp_ + t_; }                                             \
                 } else if (valid) { _Pragma("unroll") for (int i = 0; i < K; i++) if (i < a.k) {         \
                 double *mp_ = a.MV + ((size_t)(NODE) * a.k + i) * n + sl;                                \
                 const double t_ = pc * FB[i] * LB[i]; *mp_ = (first_cat ? t_ : *mp_ + t_) * (last_cat ? inv : 1.0); } } } while (0)

            /* marginal of a leaf that observes state ST_ at this site: one non-zero entry FS_ = F(ST_) (B = 1 there) */
#define UDV_OUT_M1(NODE, ST_, FS_)                                                                        \
            do { if (a.MVS) {                                                                             \
                     const double x_ = (valid ? (a.wsite ? a.wsite[sl] : 1.0) * inv * pc : 0.0) * (FS_);  \
                     double v_[K];                                                                        \
                     _Pragma("unroll") for (int i = 0; i < K; i++) v_[i] = (ST_) == i ? x_ : 0.0;         \
                     const double t_ = udv_state_sums<K>(v_, strip, threadIdx.x & 63);                   \
                     if (sum_st >= 0 && sum_st < a.k) {                                                   \
                         double *mp_ = a.MVS + ((size_t)(NODE) * a.k + sum_st) * nwv + wv;                \
                         *mp_ = first_cat ? t_ : *mp_ + t_; }                                             \
                 } else if (valid) { _Pragma("unroll") for (int i = 0; i < K; i++) if (i < a.k) {         \
                 double *mp_ = a.MV + ((size_t)(NODE) * a.k + i) * n + sl;                                \
                 const double t_ = (ST_) == i ? pc * (FS_) : 0.0; *mp_ = (first_cat ? t_ : *mp_ + t_) * (last_cat ? inv : 1.0); } } } while (0)

            /* one edge at a time; the messages of the siblings are recomputed for every edge (a binary node costs three
             * products per internal edge either way; handling both children in one visit would keep F_a, both child
             * messages and both edge-form vectors live -- 300 registers at K = 20, one wave per SIMD, measured 9 M
             * sites/s at BASELINE config 4 -- whereas this form needs ~130 and the second reading of F_a / L_b within
             * a visit comes from L2 / Infinity Cache) */
            for (int j = 0; j < deg; j++) {
                const int b = ch[4 * j], t = ch[4 * j + 1], fl = ch[4 * j + 2], bi = ch[4 * j + 3];
                if (!(fl & (UDV_WANT_D | UDV_WANT_F))) continue;
                /* forward vector of the node, with its own observation and its rescaling factor folded in (read again
                 * for every child -- from L2 the second time -- rather than kept live across the products) */
                double fe[K];
                udv_load<K>(a.FN + ((size_t)nd_int * a.C + c) * K * n, n, slc, fe);
                if (hd) {
                    double bv[K];
                    udv_gather<K>(tipc + ((size_t)a.ntips * a.nchar + a.codes[(size_t)nd * a.Spad + sg]) * K, bv);
#pragma unroll
                    for (int i = 0; i < K; i++) fe[i] *= bv[i];
                }
                if (slot >= 0) {
                    const double sc = a.SC[((size_t)slot * a.C + c) * n + slc];
#pragma unroll
                    for (int i = 0; i < K; i++) fe[i] *= sc;
                }
                for (int j2 = 0; j2 < deg; j2++) {
                    if (j2 == j) continue;
                    const int b2 = ch[4 * j2], t2 = ch[4 * j2 + 1], bi2 = ch[4 * j2 + 3];
                    double m[K];
                    if (t2 >= 0) udv_gather<K>(tipc + ((size_t)t2 * a.nchar + a.codes[(size_t)b2 * a.Spad + sg]) * K, m);
                    else {
                        double L[K];
                        UDV_CHILD_L(b2, t2, bi2, L);
                        vec_touch<K>(ms + 2 * KK);
                        vec_matvec<K>(ms, L, m);
                        ms += KK;
                        if (udv_const<K>(L, a.k)) {
#pragma unroll
                            for (int i = 0; i < K; i++) m[i] = i < a.k ? L[0] : 0.0;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < K; i++) fe[i] *= m[i];
                }
                if (DERIV && (fl & UDV_WANT_D)) {
                    double y[K];
                    if (t >= 0) udv_gather<K>(dtipc + ((size_t)t * a.nchar + a.codes[(size_t)b * a.Spad + sg]) * K, y);
                    else {
                        double L[K];
                        UDV_CHILD_L(b, t, bi, L);
                        vec_touch<K>(ms + 2 * KK);
                        vec_matvec<K>(ms, L, y);
                        ms += KK;
                        if (a.dzero && udv_const<K>(L, a.k)) {
#pragma unroll
                            for (int i = 0; i < K; i++) y[i] = 0.0;
                        }
                    }
                    double d = 0.0;
#pragma unroll
                    for (int i = 0; i < K; i++) d = fma(fe[i], y[i], d);
                    UDV_OUT_D(e0 + j, pc * d);
                }
                /* Marginal of a leaf whose whole wave observes single states: F_b o B_b has one non-zero entry, at the observed
                 * state s -- F_b(s) = sum_j P[j][s] fe[j], a K-term dot product with row s of the transposed matrix (gathered
                 * per lane, L2 resident) in the order the full product sums it (same bits), instead of the K x K product.
                 * The matrix of the stream is skipped, not consumed (its successor is still requested ahead). */
                bool leaf_done = false;
                if (MARG && t >= 0 && (fl & UDV_WANT_F) && (fl & UDV_WANT_M)) {
                    const int st = a.code_state[a.codes[(size_t)b * a.Spad + sg]];
                    if (__all(st >= 0)) {
                        leaf_done = true;
                        vec_touch<K>(ms + 2 * KK);
                        ms += KK;
                        const double *col = a.PT + ((size_t)c * a.E + e0 + j) * KK + (size_t)st * K;
                        double fs = col[0] * fe[0];
#pragma unroll
                        for (int i = 1; i < K; i++) fs = fma(col[i], fe[i], fs);
                        UDV_OUT_M1(b, st, fs);
                    }
                }
                if ((fl & UDV_WANT_F) && !leaf_done) {
                    double fb[K];
                    vec_touch<K>(ms + 2 * KK);
                    vec_matvec<K>(ms, fe, fb);
                    ms += KK;
                    if ((fl & UDV_STORE_F) && valid) udv_store<K>(a.FN + ((size_t)bi * a.C + c) * K * n, n, slc, fb);
                    if (MARG && (fl & UDV_WANT_M)) {
                        double lb[K];
                        if (t >= 0) udv_gather<K>(tipc + ((size_t)a.ntips * a.nchar + a.codes[(size_t)b * a.Spad + sg]) * K, lb);
                        else UDV_CHILD_L(b, t, bi, lb);          /* stored, or rebuilt from the tip tables (inline child) */
                        UDV_OUT_M(b, fb, lb);
                    }
                    if ((DERIV || MARG) && (fl & PLK_UP_INLINE)) {
                        /* the child's own children are leaves: their edge forms are finished here, while the child's
                         * forward vector is in registers (it is never stored, the child has no visit of its own) */
                        const PLK_AS4 int *q = vis + (-2 - t);       /* the child's inline record (plk_program.h) */
                        if (q[1]) {
                            double bv[K];
                            udv_gather<K>(tipc + ((size_t)a.ntips * a.nchar + a.codes[(size_t)b * a.Spad + sg]) * K, bv);
#pragma unroll
                            for (int i = 0; i < K; i++) fb[i] *= bv[i];
                        }
                        if (q[2] >= 0) {
                            const double sc = a.SC[((size_t)q[2] * a.C + c) * n + slc];
#pragma unroll
                            for (int i = 0; i < K; i++) fb[i] *= sc;
                        }
                        const int nl = q[3], le0 = q[4];
                        const int cd0 = a.codes[(size_t)q[5] * a.Spad + sg];
                        const int cd1 = nl == 2 ? a.codes[(size_t)q[8] * a.Spad + sg] : 0;
                        if (MARG) {
                            /* marginals of the child's leaves (each observes one state at every site: the builder inlined the
                             * child on that condition): F_leaf(s) = sum_j P_leaf[j][s] (F_b o message of the other leaf)[j] */
                            const int st0 = a.code_state[cd0], st1 = nl == 2 ? a.code_state[cd1] : 0;
                            if (q[7] & 2) {
                                double g[K];
#pragma unroll
                                for (int i = 0; i < K; i++) g[i] = fb[i];
                                if (nl == 2) {
                                    double m[K];
                                    udv_gather<K>(tipc + ((size_t)q[9] * a.nchar + cd1) * K, m);
#pragma unroll
                                    for (int i = 0; i < K; i++) g[i] *= m[i];
                                }
                                const double *col = a.PT + ((size_t)c * a.E + le0) * KK + (size_t)(st0 < 0 ? 0 : st0) * K;
                                double fs = col[0] * g[0];
#pragma unroll
                                for (int i = 1; i < K; i++) fs = fma(col[i], g[i], fs);
                                UDV_OUT_M1(q[5], st0, fs);
                            }
                            if (nl == 2 && (q[10] & 2)) {
                                double g[K], m[K];
                                udv_gather<K>(tipc + ((size_t)q[6] * a.nchar + cd0) * K, m);
#pragma unroll
                                for (int i = 0; i < K; i++) g[i] = fb[i] * m[i];
                                const double *col = a.PT + ((size_t)c * a.E + le0 + 1) * KK + (size_t)(st1 < 0 ? 0 : st1) * K;
                                double fs = col[0] * g[0];
#pragma unroll
                                for (int i = 1; i < K; i++) fs = fma(col[i], g[i], fs);
                                UDV_OUT_M1(q[8], st1, fs);
                            }
                        }
                        if (DERIV && (q[7] & 1)) {
                            double y[K], d = 0.0;
                            udv_gather<K>(dtipc + ((size_t)q[6] * a.nchar + cd0) * K, y);
                            if (nl == 2) {
                                double m[K];
                                udv_gather<K>(tipc + ((size_t)q[9] * a.nchar + cd1) * K, m);
#pragma unroll
                                for (int i = 0; i < K; i++) y[i] *= m[i];
                            }
#pragma unroll
                            for (int i = 0; i < K; i++) d = fma(fb[i], y[i], d);
                            UDV_OUT_D(le0, pc * d);
                        }
                        if (DERIV && nl == 2 && (q[10] & 1)) {
                            double y[K], m[K], d = 0.0;
                            udv_gather<K>(dtipc + ((size_t)q[9] * a.nchar + cd1) * K, y);
                            udv_gather<K>(tipc + ((size_t)q[6] * a.nchar + cd0) * K, m);
#pragma unroll
                            for (int i = 0; i < K; i++) d = fma(fb[i] * m[i], y[i], d);
                            UDV_OUT_D(le0 + 1, pc * d);
                        }
                    }
                }
            }
        }
#undef UDV_OUT_D
#undef UDV_OUT_M
#undef UDV_OUT_M1
#undef UDV_CHILD_L
    }
}

/* dtip[((c*(ntips+1) + t)*nchar + code)*K + i] = (M_e defs[code])[i] in double-double; zero for constant definition
 * rows when the matrices have zero row sums (src/util.c:338-345); slot ntips unused (zeros) */
__global__ void k_build_dtip_vec(int k, int K, int E, int ntips, int nchar, const int *__restrict__ tip_edge,
                                 const double *__restrict__ M /* [C][E][k][k] */, const double *__restrict__ defs /* [nchar][Kdef] */,
                                 int Kdef, double *__restrict__ dtip, int dzero)
{
    __shared__ int s_kind[PLK_DEF_KIND_CACHE];
    const int t = blockIdx.x, c = blockIdx.y;
    const int e = tip_edge[t];
    for (int code = threadIdx.x; code < nchar && code < PLK_DEF_KIND_CACHE; code += blockDim.x) s_kind[code] = def_row_kind(defs + (size_t)code * Kdef, k);
    __syncthreads();
    for (int idx = threadIdx.x; idx < nchar * K; idx += blockDim.x) {
        const int code = idx / K, i = idx - code * K;
        const double *d = defs + (size_t)code * Kdef;
        double out = 0.0;
        if (i < k && e >= 0) {
            const int kind = code < PLK_DEF_KIND_CACHE ? s_kind[code] : def_row_kind(d, k);
            if (!(kind == -2 && dzero)) {
                const double *row = M + ((size_t)c * E + e) * k * k + (size_t)i * k;
                if (kind >= 0) out = row[kind];             /* an observed state: column `kind` of M_e (one non-zero term, exact) */
                else {
                    dd acc = dd_make(0.0, 0.0);
                    for (int j = 0; j < k; j++) acc = dd_add(acc, dd_two_prod(row[j], d[j]));
                    out = acc.hi;
                }
            }
        }
        dtip[(((size_t)c * (ntips + 1) + t) * nchar + code) * K + i] = out;
    }
}

/* matrix stream of the up pass: MS[c][slot] = matrix kind[slot] of edge edge[slot]; kind 0: transposed P (PT layout:
 * out[j*K+i] = P[i][j]), 1: transposed M (edge form), 2: plain P (out[i*K+j] = P[i][j]); slots nstream .. nstream+2: zeros */
__global__ void k_build_up_stream(int k, int K, int E, int nstream, const int *__restrict__ kind, const int *__restrict__ edge,
                                  const double *__restrict__ P, const double *__restrict__ M, double *__restrict__ MS)
{
    const int slot = blockIdx.x, c = blockIdx.y;
    double *dst = MS + ((size_t)c * (nstream + 3) + slot) * K * K;
    const int kd = slot < nstream ? kind[slot] : -1;
    const double *src = kd < 0 ? nullptr : (kd == 1 ? M : P) + ((size_t)c * E + edge[slot < nstream ? slot : 0]) * k * k;
    for (int idx = threadIdx.x; idx < K * K; idx += blockDim.x) {
        const int r = idx / K, q = idx - r * K;
        double v = 0.0;
        if (kd >= 0 && r < k && q < k) v = kd == 2 ? src[r * k + q] : src[q * k + r];
        dst[idx] = v;
    }
}

#endif
