/*
 * plk_program.h -- the traversal program of the pruning kernels: builder, device formats and checkers.
 * Host-only C++ (no HIP types), included by plk_engine.hip and by the CPU test driver tests/progcheck_main.cpp.
 *
 * The reference walks the tree with pointers per site (src/evaluate_site_lhood.c:21-57 over the preorder of
 * src/model.h:62-67).  Here the walk is compiled once per query into a wave-uniform post-order *program* that the
 * kernels interpret for every site; the formats derived from it are what the device code indexes LDS, the matrix
 * stream, the tip tables and the register stack with.  Everything the device will dereference is therefore
 * decided on the host, and the plk_check_* functions below replay the interpreters' address arithmetic on the
 * host (same field widths, same look-ahead fetches) so that a program that would step outside a buffer is
 * refused with PLK_E_ARG before any launch.
 */
#ifndef PLK_PROGRAM_H
#define PLK_PROGRAM_H

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

/* traversal program opcodes */
enum {
    OP_TIP_SET = 0,   /* cur  = P_e * B_b         (b a leaf child)          */
    OP_TIP_MUL = 1,   /* cur *= P_e * B_b                                   */
    OP_MATVEC = 2,    /* cur  = P_e * cur         (b an internal child)     */
    OP_PUSH = 3,      /* slot[d] = cur                                      */
    OP_POPMUL = 4,    /* cur *= slot[d]                                     */
    OP_NODE_MUL = 5,  /* cur *= B_a               (internal node with data) */
    OP_SCALE = 6,     /* cur *= 2^-e, exponent accumulated (exact)          */
    OP_END = 7
};
/* handler 3 of the assembly interpreter's word format: TIP_MUL that may skip its wait (plk_fused4_asm.h) */
#define PLK_WORD_TIPMUL_NOWAIT 3u
#define PLK_WORD_SET_POPMUL 12u        /* + slot: TIP_SET and the POPMUL that follows it (two-sites-per-lane interpreter only) */
#define PLK_WORD_SET_PUSH 20u          /* + slot: TIP_SET and the PUSH that follows it */
#define PLK_WORD_MATVEC_TIPMUL 4u      /* assembly interpreter: MATVEC and the TIP_MUL that follows it, as one op word (handler slots 4 and 5) */

struct plk_op2 { int x, y; };          /* layout of HIP's int2: x = opcode | tip_slot << 8, y = node / stack slot */
struct plk_op4 { int x, y, z, w; };    /* layout of HIP's int4 */

#define PLK_TILE 256                   /* sites per workgroup of the fused k = 4 kernels */
#define PLK_FUSED_SLOTS 16             /* deepest register stack any fused kernel is compiled for */
#define PLK_LDS_LIMIT (150 * 1024)     /* dynamic LDS the fused kernels may ask for (160 KB per CU on gfx950) */

struct PlkProgram {
    std::vector<plk_op2> ops;
    std::vector<int> op_edge;          /* CSR edge per op or -1 */
    std::vector<int> tip_edge;         /* CSR edge per tip slot */
    std::vector<char> scale_node;      /* N: node vectors rescaled here (every >= 16 accumulated edges) */
    std::vector<int> obs_nodes;        /* nodes whose code rows the fused kernels stage, in order of first use */
    int slots_needed = 0;              /* Sethi-Ullman number of the tree = depth of the waiting-vector stack */
};

/*
 * Post-order program of a rooted tree in CSR form (preorder[0] = root, parents first).  Internal children are
 * visited in order of decreasing stack need, so the stack depth is the tree's Sethi-Ullman number (<= log2 of the
 * leaf count for binary trees).  node_has_data[a] != 0 for internal nodes whose observation row is not all ones.
 */
static inline void plk_program_build(int N, const int *ip, const int *ix, const int *preorder,
                                     const char *node_has_data, PlkProgram &pg)
{
    std::vector<int> need(N, 0), since(N, 0);
    pg.scale_node.assign(N, 0);
    std::vector<std::vector<int>> ichild(N); /* internal children (CSR edge idx), sorted by need desc */
    for (int u = N - 1; u >= 0; u--) {
        const int a = preorder[u];
        if (ip[a + 1] == ip[a]) continue;
        std::vector<int> &ic = ichild[a];
        int acc = 0;
        for (int idx = ip[a]; idx < ip[a + 1]; idx++) {
            const int b = ix[idx];
            acc += since[b] + 1;
            if (ip[b + 1] > ip[b]) ic.push_back(idx);
        }
        std::stable_sort(ic.begin(), ic.end(), [&](int x, int y) { return need[ix[x]] > need[ix[y]]; });
        int nd = 0;
        for (size_t i = 0; i < ic.size(); i++) nd = std::max(nd, need[ix[ic[i]]] + (i > 0 ? 1 : 0));
        need[a] = nd;
        if (acc >= 16) { pg.scale_node[a] = 1; acc = 0; }
        since[a] = acc;
    }
    const int root = preorder[0];
    pg.slots_needed = need[root];

    pg.ops.clear(); pg.op_edge.clear(); pg.tip_edge.clear(); pg.obs_nodes.clear();
    std::vector<int> obs_row(N, -1);
    auto obs = [&](int node) {
        if (obs_row[node] < 0) { obs_row[node] = (int)pg.obs_nodes.size(); pg.obs_nodes.push_back(node); }
        return obs_row[node];
    };
    auto emit = [&](int code, int tslot, int arg, int edge) {
        plk_op2 o; o.x = code | (tslot << 8); o.y = arg;
        pg.ops.push_back(o); pg.op_edge.push_back(edge);
    };
    /* iterative post-order emission; frame = (node, depth, stage) */
    struct Frame { int a, depth, stage; bool started; };
    std::vector<Frame> stk;
    stk.push_back({root, 0, 0, false});
    while (!stk.empty()) {
        Frame &f = stk.back();
        const int a = f.a;
        const std::vector<int> &ic = ichild[a];
        if (f.stage == 0) {
            f.stage = 1;
            if (!ic.empty()) { const Frame nf = {ix[ic[0]], f.depth, 0, false}; stk.push_back(nf); continue; }
        }
        if (f.stage == 1) {
            if (!ic.empty()) { emit(OP_MATVEC, 0, 0, ic[0]); f.started = true; }
            for (int idx = ip[a]; idx < ip[a + 1]; idx++) {
                const int b = ix[idx];
                if (ip[b + 1] > ip[b]) continue;
                const int t = (int)pg.tip_edge.size();
                pg.tip_edge.push_back(idx);
                emit(f.started ? OP_TIP_MUL : OP_TIP_SET, t, b, idx);
                obs(b);
                f.started = true;
            }
            f.stage = 2;
        }
        if (f.stage >= 2) {
            const int i = f.stage - 1; /* next internal child index (>= 1) */
            if (f.stage > 2) { /* returning from child i-1 */
                emit(OP_MATVEC, 0, 0, ic[i - 1]);
                emit(OP_POPMUL, 0, f.depth, -1);
            }
            if (i < (int)ic.size()) {
                emit(OP_PUSH, 0, f.depth, -1);
                f.stage++;
                const Frame nf = {ix[ic[i]], f.depth + 1, 0, false};
                stk.push_back(nf);      /* invalidates f */
                continue;
            }
            if (node_has_data[a]) { emit(OP_NODE_MUL, 0, a, -1); obs(a); }
            if (pg.scale_node[a]) emit(OP_SCALE, 0, a, -1);   /* y = the node (used by the storing down pass) */
            stk.pop_back();
        }
    }
}

static inline std::string plk_fmt(const char *fmt, long a = 0, long b = 0, long c = 0)
{
    char buf[256];
    snprintf(buf, sizeof buf, fmt, a, b, c);
    return std::string(buf);
}

/*
 * Invariants of the program itself, by abstract interpretation: every internal node's vector is completed exactly
 * once, every edge is applied exactly once, the stack is used as a stack inside [0, slots_needed), and the op
 * arguments name existing nodes / tip slots.  Returns "" when the program is well formed.
 */
static inline std::string plk_program_check(int N, const int *ip, const int *ix, const int *preorder,
                                            const char *node_has_data, const PlkProgram &pg)
{
    const int E = N - 1;
    const int nops = (int)pg.ops.size();
    if ((int)pg.op_edge.size() != nops) return "program: op_edge size";
    if ((int)pg.scale_node.size() != N) return "program: scale_node size";
    if (pg.slots_needed < 0) return "program: negative stack depth";
    std::vector<int> edge_seen(std::max(E, 1), 0), tip_seen(pg.tip_edge.size(), 0);
    std::vector<char> slot_full(std::max(pg.slots_needed, 1), 0);
    int depth = 0;
    bool have_cur = false;
    for (int pc = 0; pc < nops; pc++) {
        const int code = pg.ops[pc].x & 0xff, t = pg.ops[pc].x >> 8, y = pg.ops[pc].y, e = pg.op_edge[pc];
        switch (code) {
        case OP_TIP_SET: case OP_TIP_MUL:
            if (t < 0 || t >= (int)pg.tip_edge.size() || tip_seen[t]++) return plk_fmt("program: op %ld: bad tip slot %ld", pc, t);
            if (e != pg.tip_edge[t] || e < 0 || e >= E || ix[e] != y) return plk_fmt("program: op %ld: tip edge mismatch", pc);
            if (ip[y + 1] != ip[y]) return plk_fmt("program: op %ld: tip op on an internal node", pc);
            if (edge_seen[e]++) return plk_fmt("program: edge %ld applied twice", e);
            if ((code == OP_TIP_SET) == have_cur) return plk_fmt("program: op %ld: TIP_SET/TIP_MUL does not match the state", pc);
            have_cur = true;
            break;
        case OP_MATVEC:
            if (e < 0 || e >= E || edge_seen[e]++) return plk_fmt("program: op %ld: bad or repeated edge %ld", pc, e);
            if (ip[ix[e] + 1] == ip[ix[e]]) return plk_fmt("program: op %ld: MATVEC on a leaf edge", pc);
            if (!have_cur) return plk_fmt("program: op %ld: MATVEC without a vector", pc);
            break;
        case OP_PUSH:
            if (y != depth || y >= pg.slots_needed || slot_full[y] || !have_cur) return plk_fmt("program: op %ld: bad PUSH to slot %ld (depth %ld)", pc, y, depth);
            slot_full[y] = 1; depth++; have_cur = false;
            break;
        case OP_POPMUL:
            if (y != depth - 1 || y < 0 || !slot_full[y] || !have_cur) return plk_fmt("program: op %ld: bad POPMUL of slot %ld (depth %ld)", pc, y, depth);
            slot_full[y] = 0; depth--;
            break;
        case OP_NODE_MUL:
            if (y < 0 || y >= N || ip[y + 1] == ip[y] || !node_has_data[y] || !have_cur) return plk_fmt("program: op %ld: bad NODE_MUL", pc);
            break;
        case OP_SCALE:
            if (y < 0 || y >= N || !pg.scale_node[y] || !have_cur) return plk_fmt("program: op %ld: bad SCALE", pc);
            break;
        default:
            return plk_fmt("program: op %ld: unknown opcode %ld", pc, code);
        }
    }
    if (depth != 0) return "program: stack not empty at the end";
    if (E > 0 && !have_cur) return "program: no root vector";
    for (int e = 0; e < E; e++) if (edge_seen[e] != 1) return plk_fmt("program: edge %ld applied %ld times", e, edge_seen[e]);
    for (size_t r = 0; r < pg.obs_nodes.size(); r++)
        if (pg.obs_nodes[r] < 0 || pg.obs_nodes[r] >= N) return "program: staged row names a node out of range";
    (void)preorder;
    return "";
}

/* ------------------------------------------------------------------------------------------------------------ */
/* fused k = 4 kernels: int4 program of the C++ interpreter, 32-bit op words of the assembly interpreter           */
/* ------------------------------------------------------------------------------------------------------------ */

struct PlkFused {
    std::vector<plk_op4> fops;         /* x = opcode | tip<<8, y = row / slot, z = next observation row, w = matrix index */
    std::vector<int> mat_edge;         /* CSR edge per matrix of the compact stream (the kernels add one spare) */
    std::vector<unsigned> words;       /* blocks of 8, END padded, one spare block */
    int first_row = 0;
    int asm_first_tip = 0, asm_first_row = 0, asm_second_row = 0;
    bool asm_ok = false;               /* the assembly interpreter's field widths and stack depth suffice */
};

/* the assembly interpreter's op-word fields: 8 stack slots, 11 bits of tip slot, 16 bits of staged row */
static inline bool plk_fused_asm_ok(const PlkProgram &pg)
{
    return pg.slots_needed <= 8 && pg.tip_edge.size() + 1 < 2048 && pg.obs_nodes.size() < 65536;
}

static inline void plk_fused_build(int N, const PlkProgram &pg, PlkFused &fu)
{
    const int nops = (int)pg.ops.size();
    std::vector<int> row(N, -1);
    for (size_t r = 0; r < pg.obs_nodes.size(); r++) row[pg.obs_nodes[r]] = (int)r;
    /* C++ interpreter: ops are executed in pairs and fetched one pair ahead: even count + one spare pair */
    const int npad = ((nops + 1) / 2) * 2 + 2;
    fu.fops.assign(npad, plk_op4{OP_END, 0, 0, 0});
    fu.mat_edge.clear();
    int next_row = 0;
    for (int pc = nops - 1; pc >= 0; pc--) {
        const plk_op2 o = pg.ops[pc];
        const int code = o.x & 0xff;
        plk_op4 f = {o.x, o.y, next_row, 0};
        if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) { f.y = row[o.y]; next_row = f.y; }
        fu.fops[pc] = f;
    }
    fu.first_row = next_row;
    for (int pc = 0; pc < nops; pc++)
        if ((fu.fops[pc].x & 0xff) == OP_MATVEC) { fu.fops[pc].w = (int)fu.mat_edge.size(); fu.mat_edge.push_back(pg.op_edge[pc]); }
    /* assembly interpreter: 32-bit words {handler index | y<<5 | z<<16} in blocks of 8 (handler index = opcode, or
     * 8 + slot for PUSH, 16 + slot for POPMUL: the interpreter jumps to base + 256 * index); an observation op carries
     * the tip slot of the NEXT observation op (y) and the code row of the one after next (z): the prefetch chain of
     * plk_fused4_asm.h.  An internal node's own data is an observation on the pseudo tip slot ntips. */
    const int ntips = (int)pg.tip_edge.size();
    std::vector<int> obs_t, obs_row;
    for (int pc = 0; pc < nops; pc++) {
        const int code = fu.fops[pc].x & 0xff;
        if (code == OP_TIP_SET || code == OP_TIP_MUL) { obs_t.push_back(fu.fops[pc].x >> 8); obs_row.push_back(fu.fops[pc].y); }
        else if (code == OP_NODE_MUL) { obs_t.push_back(ntips); obs_row.push_back(fu.fops[pc].y); }
    }
    const int nwords = ((nops + 1 + 7) / 8) * 8 + 8;
    fu.words.assign(nwords, (unsigned)OP_END);
    size_t oi = 0, nw = 0;
    bool matvec_since_obs = false;   /* a MATVEC (full wait) ran since the last observation op */
    for (int pc = 0; pc < nops; pc++) {
        const int code = fu.fops[pc].x & 0xff;
        unsigned wv = (unsigned)code;
        auto obs_fields = [&]() {
            const unsigned tn = oi + 1 < obs_t.size() ? (unsigned)obs_t[oi + 1] : 0u;
            const unsigned rn = oi + 2 < obs_row.size() ? (unsigned)obs_row[oi + 2] : 0u;
            return (tn << 5) | (rn << 16);
        };
        if (code == OP_MATVEC) {
            const int nx = pc + 1 < nops ? (fu.fops[pc + 1].x & 0xff) : OP_END;
            if ((nx == OP_TIP_MUL || nx == OP_NODE_MUL) && oi > 0) {
                /* MATVEC followed by TIP_MUL (an internal child, then a leaf child or the node's own data): one word,
                 * handler 3; the product's wait covers the prefetched tip value */
                fu.words[nw++] = PLK_WORD_MATVEC_TIPMUL | obs_fields();
                oi++;
                matvec_since_obs = false;
                pc++;
                continue;
            }
            matvec_since_obs = true;
            if ((nx == OP_PUSH || nx == OP_POPMUL) && pg.slots_needed <= 4) {
                /* MATVEC followed by PUSH / POPMUL (a node with two internal children): one word, handlers 24 + d /
                 * 28 + d of the 4-slot interpreter */
                fu.words[nw++] = (nx == OP_PUSH ? 24u : 28u) + (unsigned)fu.fops[pc + 1].y;
                pc++;
                continue;
            }
        }
        if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
            const unsigned oc = code == OP_TIP_SET ? (unsigned)OP_TIP_SET
                                                   : (matvec_since_obs && oi > 0 ? PLK_WORD_TIPMUL_NOWAIT : (unsigned)OP_TIP_MUL);
            wv = oc | obs_fields();
            matvec_since_obs = false;
            oi++;
        } else if (code == OP_PUSH || code == OP_POPMUL) {
            wv = (code == OP_PUSH ? 8u : 16u) + (unsigned)fu.fops[pc].y;
        }
        fu.words[nw++] = wv;
    }
    fu.asm_first_tip = obs_t.empty() ? 0 : obs_t[0];
    fu.asm_first_row = obs_row.empty() ? 0 : obs_row[0];
    fu.asm_second_row = obs_row.size() > 1 ? obs_row[1] : 0;
    fu.asm_ok = plk_fused_asm_ok(pg);
}

/* dynamic LDS of the fused ll kernels: tip tables of ncat categories (ntips + the pseudo slot each) + staged code rows */
static inline size_t plk_fused_lds_bytes(const PlkProgram &pg, int nchar, int bytes_per_row, int ncat = 1)
{
    return (size_t)ncat * (pg.tip_edge.size() + 1) * nchar * 4 * sizeof(double) + pg.obs_nodes.size() * (size_t)bytes_per_row;
}

/* the observation sequence of the program: (tip slot incl. pseudo slot, staged row) per observation op */
static inline void plk_obs_sequence(int N, const PlkProgram &pg, std::vector<int> &slot, std::vector<int> &rowv, std::vector<int> &opc)
{
    std::vector<int> row(N, -1);
    for (size_t r = 0; r < pg.obs_nodes.size(); r++) row[pg.obs_nodes[r]] = (int)r;
    slot.clear(); rowv.clear(); opc.clear();
    for (size_t pc = 0; pc < pg.ops.size(); pc++) {
        const int code = pg.ops[pc].x & 0xff;
        if (code == OP_TIP_SET || code == OP_TIP_MUL) { slot.push_back(pg.ops[pc].x >> 8); rowv.push_back(row[pg.ops[pc].y]); opc.push_back(code); }
        else if (code == OP_NODE_MUL) { slot.push_back((int)pg.tip_edge.size()); rowv.push_back(row[pg.ops[pc].y]); opc.push_back(code); }
    }
}

/*
 * Replays k_ll_fused4_asm<D> (plk_fused4_asm.h) on the host: the block-ahead word fetch, the matrix stream that is
 * always one matrix ahead, the LDS addresses of the tip-value prefetch chain (for the largest code, nchar - 1) and of
 * the code bytes, the 13-bit / 16-bit fields, the AGPR slot numbers -- and checks that the values the chain delivers
 * to each observation op are the ones the program asks for.  pack4: two codes per staged byte (nchar <= 16).
 */
static inline std::string plk_fused_check_asm(int N, const PlkProgram &pg, const PlkFused &fu, int nchar, int D, int pack4,
                                              size_t lds_bytes_launched, int ncat_lds = 1)
{
    const int nops = (int)pg.ops.size(), ntips1 = (int)pg.tip_edge.size() + 1, nobs = (int)pg.obs_nodes.size();
    const int nmat = (int)fu.mat_edge.size();
    if (nchar < 1 || nchar > 256) return "asm program: nchar out of range";
    if (pack4 && nchar > 16) return "asm program: 4-bit codes need nchar <= 16";
    if (D != 4 && D != 8) return "asm program: stack depth variant";
    if (pg.slots_needed > D) return "asm program: the tree needs a deeper register stack";
    if (fu.words.size() % 8 != 0 || (int)fu.words.size() < ((nops + 1 + 7) / 8) * 8 + 8) return "asm program: word buffer too short";
    const size_t row_bytes = pack4 ? PLK_TILE / 2 : PLK_TILE;
    const size_t tip_bytes = (size_t)ntips1 * nchar * 32, code_bytes = (size_t)nobs * row_bytes;
    if (ncat_lds < 1) return "asm program: categories per pass";
    if ((size_t)ncat_lds * tip_bytes + code_bytes > lds_bytes_launched) return "asm program: LDS image larger than the launch's dynamic LDS";
    if (lds_bytes_launched > PLK_LDS_LIMIT) return "asm program: dynamic LDS above the limit";
    if (nobs < 1) return "asm program: no observation rows";
    std::vector<int> slot, rowv, opc;
    plk_obs_sequence(N, pg, slot, rowv, opc);
    auto tip_ok = [&](long t) { return t >= 0 && t < ntips1 && (size_t)t * nchar * 32 + (size_t)(nchar - 1) * 32 + 32 <= tip_bytes; };
    auto row_ok = [&](long r) { return r >= 0 && r < nobs && (size_t)r * row_bytes + (row_bytes - 1) < code_bytes; };
    if (!tip_ok(fu.asm_first_tip) || !row_ok(fu.asm_first_row) || !row_ok(fu.asm_second_row)) return "asm program: prologue prefetch out of range";
    /* prefetch chain state: value in flight = (cur_tip, cur_row), code in flight = next_row */
    long cur_tip = fu.asm_first_tip, cur_row = fu.asm_first_row, next_row = fu.asm_second_row;
    size_t oi = 0;
    int mi = 0;                       /* matrix currently loaded; the kernel loads mi + 1 after each MATVEC */
    bool waited = true;               /* lgkmcnt(0) seen since the in-flight value was requested (prologue waits) */
    std::vector<char> full(D, 0);
    const size_t nblocks = fu.words.size() / 8;
    size_t pc = 0;                    /* op of the program the next word stands for (a pair word stands for two) */
    /* one observation op of the chain: the word's fields name the next observation's slot and the row after next */
    auto observation = [&](size_t wi, unsigned y, unsigned z, bool is_set, bool no_wait) -> std::string {
        if ((int)pc >= nops) return "asm program: op beyond the program";
        const int code = pg.ops[pc].x & 0xff;
        if (code != OP_TIP_SET && code != OP_TIP_MUL && code != OP_NODE_MUL) return plk_fmt("asm program: word %ld is not an observation op", (long)wi);
        if (is_set != (code == OP_TIP_SET)) return plk_fmt("asm program: word %ld: SET/MUL mismatch", (long)wi);
        if (no_wait && !waited) return plk_fmt("asm program: word %ld skips a wait it needs", (long)wi);
        if (oi >= slot.size() || cur_tip != slot[oi] || cur_row != rowv[oi]) return plk_fmt("asm program: prefetch chain delivers the wrong observation at word %ld", (long)wi);
        if (!tip_ok(y) || !row_ok(z)) return plk_fmt("asm program: word %ld prefetches out of range (slot %ld, row %ld)", (long)wi, y, z);
        cur_tip = y; cur_row = next_row; next_row = z;
        if (oi + 1 < slot.size() && (cur_tip != slot[oi + 1] || cur_row != rowv[oi + 1])) return plk_fmt("asm program: chain after word %ld", (long)wi);
        oi++;
        waited = false;
        pc++;
        return "";
    };
    auto product = [&](size_t wi) -> std::string {
        if ((int)pc >= nops) return "asm program: op beyond the program";
        if ((pg.ops[pc].x & 0xff) != OP_MATVEC) return plk_fmt("asm program: word %ld is not the program's op", (long)wi);
        if (mi >= nmat || fu.mat_edge[mi] != pg.op_edge[pc]) return plk_fmt("asm program: matrix %ld is not the op's edge", mi);
        mi++;                                   /* loads matrix mi (<= nmat: the spare) */
        waited = true;
        pc++;
        return "";
    };
    for (size_t b = 0;; b++) {
        if (b + 1 >= nblocks) return "asm program: ran past the spare block (no END)";
        /* s_load_dwordx8 of block b + 1 is issued here: in range by the test above */
        for (int i = 0; i < 8; i++) {
            const size_t wi = b * 8 + i;
            const unsigned w = fu.words[wi];
            /* the kernel jumps to handler (w & 31): 0..7 = opcode (3 = TIP_MUL without wait, 4 = MATVEC + TIP_MUL, occupying slots 4 and 5),
             * 8 + d = PUSH slot d, 16 + d = POPMUL slot d, 24 + d / 28 + d = MATVEC then PUSH / POPMUL slot d (4-slot
             * interpreter only) */
            const unsigned hidx = w & 31, z = w >> 16;
            if (hidx == 5 || (hidx >= 24 && D != 4)) return plk_fmt("asm program: word %ld jumps to an empty handler slot", (long)wi);
            const unsigned y = hidx >= 24 ? (hidx & 3) : hidx >= 8 ? (hidx & 7) : (w >> 5) & 0x7ff;
            std::string bad;
            if (hidx == OP_END) {
                if ((int)pc != nops) return plk_fmt("asm program: END at word %ld after %ld of the program's ops", (long)wi, (long)pc);
                if (oi != slot.size()) return "asm program: observation ops missing";
                if (mi != nmat) return "asm program: matrix stream not consumed";
                return "";
            } else if (hidx == OP_MATVEC) bad = product(wi);
            else if (hidx == PLK_WORD_MATVEC_TIPMUL) { bad = product(wi); if (bad.empty()) bad = observation(wi, y, z, false, true); }
            else if (hidx == OP_TIP_SET || hidx == OP_TIP_MUL || hidx == PLK_WORD_TIPMUL_NOWAIT) bad = observation(wi, y, z, hidx == OP_TIP_SET, hidx == PLK_WORD_TIPMUL_NOWAIT);
            else if (hidx == OP_SCALE) { if ((int)pc >= nops || (pg.ops[pc].x & 0xff) != OP_SCALE) bad = plk_fmt("asm program: word %ld is not the program's op", (long)wi); pc++; }
            else {
                if (hidx >= 24) { bad = product(wi); if (!bad.empty()) return bad; }         /* MATVEC + PUSH / POPMUL */
                const bool push = hidx >= 24 ? hidx < 28 : hidx < 16;
                if ((int)pc >= nops || (pg.ops[pc].x & 0xff) != (push ? OP_PUSH : OP_POPMUL) || (int)y >= D || (int)y != pg.ops[pc].y || (full[y] != 0) == push)
                    bad = plk_fmt(push ? "asm program: bad PUSH at word %ld" : "asm program: bad POPMUL at word %ld", (long)wi);
                else full[y] = push;
                pc++;
            }
            if (!bad.empty()) return bad;
        }
        waited = true;                /* s_waitcnt lgkmcnt(0) at the end of every block */
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* pair-table interpreter k_ll_fused4_asm_pt (plk_fused4_asm.h): 512-site tiles, byte rows, cherries as look-ups     */
/* ------------------------------------------------------------------------------------------------------------ */

/* dynamic LDS the pair-table kernels may ask for, by sites per tile: one workgroup per CU (1024 sites on 1024 lanes, or
 * 1024 / 1536 sites on 512 / 768 lanes with two sites each), or two 512-site workgroups */
static inline size_t plk_pt_lds_limit(int tile) { return tile >= 1024 ? (size_t)156 * 1024 : (size_t)78 * 1024; }
static inline bool plk_pt_tile_ok(int tile) { return tile == 512 || tile == 1024 || tile == 1536; }

struct PlkFusedPT {
    std::vector<unsigned> words;       /* blocks of 8, END padded, one spare block (format of PlkFused::words; the y field
                                          counts table UNITS of nchar * 32 bytes) */
    std::vector<int> mat_edge;         /* CSR edge per matrix of the stream (the kernel adds one spare) */
    std::vector<int> row_node, row_node2;   /* staged rows: node | second node of a pair row or -1 */
    /* tables of one category, in unit order: a single leaf (1 unit: P_e defs), a pair (nchar units:
     * P_a (P_b defs[i] o P_c defs[j]) at code i * nchar + j) or the pseudo table (1 unit: the raw definitions) */
    std::vector<int> tab_unit, tab_edge, tab_eb, tab_ec;   /* first unit | leaf edge, edge above the cherry or -1 | pair: leaf edges, else -1 */
    int units = 0, npairs = 0;
    int first_unit = 0, first_row = 0, second_row = 0;
    /* the program after folding the cherries, op by op (what the words encode; the vector kernel's int4 program is made from it) */
    struct VOp { int code, unit, row, d, edge; };
    std::vector<VOp> vops;
};

/* the node whose CSR child range holds edge idx */
static inline int plk_edge_parent(int N, const int *ip, int idx)
{
    int lo = 0, hi = N - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (ip[mid] <= idx) lo = mid; else hi = mid - 1; }
    return lo;
}

/* does the program continue at pc with TIP_SET(b), TIP_MUL(c), MATVEC(edge into a) for a node a whose only children
 * are the leaves b and c? */
static inline bool plk_pair_at(int N, const int *ip, const int *ix, const PlkProgram &pg, size_t pc)
{
    if (pc + 2 >= pg.ops.size()) return false;
    if ((pg.ops[pc].x & 0xff) != OP_TIP_SET || (pg.ops[pc + 1].x & 0xff) != OP_TIP_MUL || (pg.ops[pc + 2].x & 0xff) != OP_MATVEC) return false;
    const int eb = pg.op_edge[pc], ec = pg.op_edge[pc + 1], ea = pg.op_edge[pc + 2];
    const int a = plk_edge_parent(N, ip, eb);
    if (ip[a + 1] - ip[a] != 2 || plk_edge_parent(N, ip, ec) != a || eb == ec) return false;
    return ea >= 0 && ix[ea] == a;
}

/* max_pairs: how many cherries may become tables (LDS budget); nchar * nchar <= 256 is the caller's business */
static inline void plk_fused_pt_build(int N, const int *ip, const int *ix, const PlkProgram &pg, int nchar, int max_pairs, PlkFusedPT &fu,
                                      bool fuse_set = false)
{
    const int nops = (int)pg.ops.size();
    typedef PlkFusedPT::VOp VOp;
    std::vector<VOp> &v = fu.vops;
    v.clear();
    fu.mat_edge.clear(); fu.row_node.clear(); fu.row_node2.clear();
    fu.tab_unit.clear(); fu.tab_edge.clear(); fu.tab_eb.clear(); fu.tab_ec.clear();
    fu.units = 0; fu.npairs = 0;
    std::vector<int> row_of(N, -1);
    auto single_row = [&](int node) {
        if (row_of[node] < 0) { row_of[node] = (int)fu.row_node.size(); fu.row_node.push_back(node); fu.row_node2.push_back(-1); }
        return row_of[node];
    };
    auto table = [&](int edge, int eb, int ec, int nunits) {
        fu.tab_unit.push_back(fu.units); fu.tab_edge.push_back(edge); fu.tab_eb.push_back(eb); fu.tab_ec.push_back(ec);
        const int u = fu.units; fu.units += nunits; return u;
    };
    int pseudo_unit = -1;
    for (int pc = 0; pc < nops; pc++) {
        const int code = pg.ops[pc].x & 0xff;
        if (code == OP_TIP_SET && fu.npairs < max_pairs && plk_pair_at(N, ip, ix, pg, (size_t)pc)) {
            const int row = (int)fu.row_node.size();
            fu.row_node.push_back(pg.ops[pc].y); fu.row_node2.push_back(pg.ops[pc + 1].y);
            v.push_back(VOp{OP_TIP_SET, table(pg.op_edge[pc + 2], pg.op_edge[pc], pg.op_edge[pc + 1], nchar), row, 0, -1});
            fu.npairs++;
            pc += 2;
        } else if (code == OP_TIP_SET || code == OP_TIP_MUL) {
            v.push_back(VOp{code, table(pg.op_edge[pc], -1, -1, 1), single_row(pg.ops[pc].y), 0, -1});
        } else if (code == OP_NODE_MUL) {
            if (pseudo_unit < 0) pseudo_unit = table(-1, -1, -1, 1);
            v.push_back(VOp{OP_TIP_MUL, pseudo_unit, single_row(pg.ops[pc].y), 0, -1});
        } else if (code == OP_MATVEC) {
            fu.mat_edge.push_back(pg.op_edge[pc]);
            v.push_back(VOp{OP_MATVEC, 0, 0, 0, pg.op_edge[pc]});
        } else {
            v.push_back(VOp{code, 0, 0, pg.ops[pc].y, -1});
        }
    }
    if (pseudo_unit < 0) table(-1, -1, -1, 1);      /* always present: keeps the image non-empty and the layout regular */
    std::vector<int> ou, orow;
    for (const VOp &o : v) if (o.code == OP_TIP_SET || o.code == OP_TIP_MUL) { ou.push_back(o.unit); orow.push_back(o.row); }
    const int nv = (int)v.size();
    fu.words.assign(((nv + 1 + 7) / 8) * 8 + 8, (unsigned)OP_END);
    size_t oi = 0, nw = 0;
    bool matvec_since_obs = false;
    for (int i = 0; i < nv; i++) {
        const int code = v[i].code;
        unsigned wv = (unsigned)code;
        auto obs_fields = [&]() {
            /* cyclic: the last observations request the first ones again, so that every request of the chain pairs a
             * table with a row of its own kind (the values are dropped; the next category's prologue asks again) */
            const unsigned un = (unsigned)ou[(oi + 1) % ou.size()];
            const unsigned rn = (unsigned)orow[(oi + 2) % orow.size()];
            return (un << 5) | (rn << 16);
        };
        if (code == OP_MATVEC) {
            const int nx = i + 1 < nv ? v[i + 1].code : OP_END;
            if (nx == OP_TIP_MUL && oi > 0) {
                fu.words[nw++] = PLK_WORD_MATVEC_TIPMUL | obs_fields();
                oi++; matvec_since_obs = false; i++;
                continue;
            }
            matvec_since_obs = true;
            if ((nx == OP_PUSH || nx == OP_POPMUL) && pg.slots_needed <= 4) {
                fu.words[nw++] = (nx == OP_PUSH ? 24u : 28u) + (unsigned)v[i + 1].d;
                i++;
                continue;
            }
        }
        if (code == OP_TIP_SET || code == OP_TIP_MUL) {
            unsigned oc = code == OP_TIP_SET ? (unsigned)OP_TIP_SET
                                             : (matvec_since_obs && oi > 0 ? PLK_WORD_TIPMUL_NOWAIT : (unsigned)OP_TIP_MUL);
            const int nx = i + 1 < nv ? v[i + 1].code : OP_END;
            if (fuse_set && code == OP_TIP_SET && (nx == OP_POPMUL || nx == OP_PUSH)) {
                /* TIP_SET followed by POPMUL / PUSH (a look-up as the last or as an earlier internal child): one word, handlers
                 * 12 + d / 20 + d of the two-sites-per-lane interpreter; the value goes straight into the product / the slot */
                oc = (nx == OP_POPMUL ? PLK_WORD_SET_POPMUL : PLK_WORD_SET_PUSH) + (unsigned)v[i + 1].d;
                i++;
            }
            wv = oc | obs_fields();
            matvec_since_obs = false;
            oi++;
        } else if (code == OP_PUSH || code == OP_POPMUL) {
            wv = (code == OP_PUSH ? 8u : 16u) + (unsigned)v[i].d;
        }
        fu.words[nw++] = wv;
    }
    fu.first_unit = ou.empty() ? 0 : ou[0];
    fu.first_row = orow.empty() ? 0 : orow[0];
    fu.second_row = orow.size() > 1 ? orow[1] : 0;
}

static inline size_t plk_fused_pt_lds_bytes(const PlkFusedPT &fu, int nchar, int tile)
{
    return (size_t)fu.units * nchar * 32 + fu.row_node.size() * (size_t)tile;
}

/*
 * Replays k_ll_fused4_asm_pt on the host against the PROGRAM (not against the builder's intermediate list): block-ahead
 * word fetch, matrix stream one ahead, LDS addresses of the value / code prefetch chain for the largest code of every
 * table, field widths, stack slots; every observation word must deliver the table and the staged row of the program's
 * next leaf op -- or, for a pair table, stand for exactly TIP_SET(b), TIP_MUL(c), MATVEC(edge above) of one cherry.
 */
static inline std::string plk_fused_check_pt(int N, const int *ip, const int *ix, const PlkProgram &pg, const PlkFusedPT &fu, int nchar,
                                             int tile, size_t lds_bytes_launched, bool fused_set_words = false)
{
    const int nops = (int)pg.ops.size(), nrows = (int)fu.row_node.size(), nmat = (int)fu.mat_edge.size(), ntab = (int)fu.tab_unit.size();
    if (nchar < 1 || nchar > 16) return "pt program: pair tables need nchar <= 16";
    if (pg.slots_needed > 4) return "pt program: the tree needs a deeper register stack";
    if (fu.words.size() % 8 != 0 || fu.words.size() < 16) return "pt program: word buffer";
    if ((int)fu.row_node2.size() != nrows || (int)fu.tab_edge.size() != ntab || (int)fu.tab_eb.size() != ntab || (int)fu.tab_ec.size() != ntab) return "pt program: table sizes";
    const size_t tip_bytes = (size_t)fu.units * nchar * 32, code_bytes = (size_t)nrows * tile;
    if (tip_bytes + code_bytes > lds_bytes_launched) return "pt program: LDS image larger than the launch's dynamic LDS";
    if (!plk_pt_tile_ok(tile) || lds_bytes_launched > plk_pt_lds_limit(tile)) return "pt program: dynamic LDS above the limit";
    if (nrows < 1 || fu.units < 1 || fu.units >= 2048 || nrows >= 65536) return "pt program: field widths";
    for (int r = 0; r < nrows; r++)
        if (fu.row_node[r] < 0 || fu.row_node[r] >= N || fu.row_node2[r] < -1 || fu.row_node2[r] >= N) return "pt program: staged row names a node out of range";
    /* tables tile the unit range without gaps */
    std::vector<int> table_at(fu.units, -1);
    int expect = 0;
    for (int t = 0; t < ntab; t++) {
        const int nu = fu.tab_eb[t] >= 0 ? nchar : 1;
        if (fu.tab_unit[t] != expect || expect + nu > fu.units) return "pt program: table layout";
        table_at[expect] = t;
        expect += nu;
    }
    if (expect != fu.units) return "pt program: table layout";
    auto unit_ok = [&](long u) {
        if (u < 0 || u >= fu.units || table_at[u] < 0) return false;
        const long ncodes = fu.tab_eb[table_at[u]] >= 0 ? (long)nchar * nchar : nchar;
        return (size_t)u * nchar * 32 + (size_t)ncodes * 32 <= tip_bytes;
    };
    auto row_ok = [&](long r) { return r >= 0 && r < nrows && (size_t)r * tile + (tile - 1) < code_bytes; };
    if (!unit_ok(fu.first_unit) || !row_ok(fu.first_row) || !row_ok(fu.second_row)) return "pt program: prologue prefetch out of range";
    if (fu.row_node2[fu.first_row] >= 0 && fu.tab_eb[table_at[fu.first_unit]] < 0) return "pt program: the prologue indexes a one-unit table with a pair row's code";
    long cur_unit = fu.first_unit, cur_row = fu.first_row, next_row = fu.second_row;
    int mi = 0;
    bool waited = true, any_obs = false;
    std::vector<char> full(4, 0), tab_used(ntab, 0);
    const size_t nblocks = fu.words.size() / 8;
    size_t pc = 0;
    auto code_at = [&](size_t q) { return q < (size_t)nops ? (pg.ops[q].x & 0xff) : (int)OP_END; };
    auto observation = [&](size_t wi, unsigned y, unsigned z, bool is_set, bool no_wait) -> std::string {
        if (no_wait && !waited) return plk_fmt("pt program: word %ld skips a wait it needs", (long)wi);
        if (!unit_ok(cur_unit) || !row_ok(cur_row)) return plk_fmt("pt program: word %ld consumes an observation out of range", (long)wi);
        const int t = table_at[cur_unit];
        if (tab_used[t]++ && fu.tab_edge[t] >= 0) return plk_fmt("pt program: table %ld used twice", (long)t);
        const int code = code_at(pc);
        if (fu.tab_eb[t] >= 0) {
            /* a pair: the program must continue with the cherry's three ops */
            if (!is_set || !plk_pair_at(N, ip, ix, pg, pc)) return plk_fmt("pt program: word %ld is a pair look-up where the program has no cherry", (long)wi);
            if (fu.tab_eb[t] != pg.op_edge[pc] || fu.tab_ec[t] != pg.op_edge[pc + 1] || fu.tab_edge[t] != pg.op_edge[pc + 2]) return plk_fmt("pt program: pair table %ld holds other edges than the cherry at word %ld", (long)t, (long)wi);
            if (fu.row_node[cur_row] != pg.ops[pc].y || fu.row_node2[cur_row] != pg.ops[pc + 1].y) return plk_fmt("pt program: pair row of word %ld", (long)wi);
            pc += 3;
        } else {
            if (code != OP_TIP_SET && code != OP_TIP_MUL && code != OP_NODE_MUL) return plk_fmt("pt program: word %ld is not an observation op", (long)wi);
            if (is_set != (code == OP_TIP_SET)) return plk_fmt("pt program: word %ld: SET/MUL mismatch", (long)wi);
            if (fu.row_node[cur_row] != pg.ops[pc].y || fu.row_node2[cur_row] != -1) return plk_fmt("pt program: word %ld reads the wrong staged row", (long)wi);
            if (fu.tab_edge[t] != (code == OP_NODE_MUL ? -1 : pg.op_edge[pc])) return plk_fmt("pt program: word %ld reads the wrong table", (long)wi);
            pc++;
        }
        /* what the word requests: the value of the next observation = table y at the code that is in flight (row next_row),
         * and the code of the observation after that (row z).  A pair row's codes reach nchar^2 - 1: only a pair table holds them. */
        if (!unit_ok(y) || !row_ok(z) || !row_ok(next_row)) return plk_fmt("pt program: word %ld prefetches out of range (unit %ld, row %ld)", (long)wi, y, z);
        if (fu.row_node2[next_row] >= 0 && fu.tab_eb[table_at[y]] < 0) return plk_fmt("pt program: word %ld indexes a one-unit table with a pair row's code", (long)wi);
        cur_unit = y; cur_row = next_row; next_row = z;
        waited = false; any_obs = true;
        return "";
    };
    auto product = [&](size_t wi) -> std::string {
        if (code_at(pc) != OP_MATVEC) return plk_fmt("pt program: word %ld is not the program's op", (long)wi);
        if (mi >= nmat || fu.mat_edge[mi] != pg.op_edge[pc]) return plk_fmt("pt program: matrix %ld is not the op's edge", mi);
        mi++; waited = true; pc++;
        return "";
    };
    for (size_t b = 0;; b++) {
        if (b + 1 >= nblocks) return "pt program: ran past the spare block (no END)";
        for (int i = 0; i < 8; i++) {
            const size_t wi = b * 8 + i;
            const unsigned w = fu.words[wi];
            const unsigned hidx = w & 31, z = w >> 16;
            if (hidx == 5) return plk_fmt("pt program: word %ld jumps to an empty handler slot", (long)wi);
            const bool setstack = (hidx >= PLK_WORD_SET_POPMUL && hidx < PLK_WORD_SET_POPMUL + 4) || (hidx >= PLK_WORD_SET_PUSH && hidx < PLK_WORD_SET_PUSH + 4);
            if (setstack && !fused_set_words) return plk_fmt("pt program: word %ld jumps to an empty handler slot", (long)wi);
            const unsigned y = setstack ? (w >> 5) & 0x7ff : hidx >= 24 ? (hidx & 3) : hidx >= 8 ? (hidx & 7) : (w >> 5) & 0x7ff;
            std::string bad;
            if (setstack) {
                /* TIP_SET (single or pair look-up), then POPMUL / PUSH of slot hidx & 3 */
                bad = observation(wi, y, z, true, false);
                if (!bad.empty()) return bad;
                const bool push = hidx >= PLK_WORD_SET_PUSH;
                const unsigned d = hidx & 3;
                if (code_at(pc) != (push ? OP_PUSH : OP_POPMUL) || (int)d != pg.ops[pc].y || (full[d] != 0) == push)
                    return plk_fmt(push ? "pt program: bad SET + PUSH at word %ld" : "pt program: bad SET + POPMUL at word %ld", (long)wi);
                full[d] = push;
                pc++;
                continue;
            }
            if (hidx == OP_END) {
                if ((int)pc != nops) return plk_fmt("pt program: END at word %ld after %ld of the program's ops", (long)wi, (long)pc);
                if (mi != nmat) return "pt program: matrix stream not consumed";
                for (int t = 0; t < ntab; t++) if (fu.tab_edge[t] >= 0 && tab_used[t] != 1) return plk_fmt("pt program: table %ld is never read", (long)t);
                (void)any_obs;
                return "";
            } else if (hidx == OP_MATVEC) bad = product(wi);
            else if (hidx == PLK_WORD_MATVEC_TIPMUL) { bad = product(wi); if (bad.empty()) bad = observation(wi, y, z, false, true); }
            else if (hidx == OP_TIP_SET || hidx == OP_TIP_MUL || hidx == PLK_WORD_TIPMUL_NOWAIT) bad = observation(wi, y, z, hidx == OP_TIP_SET, hidx == PLK_WORD_TIPMUL_NOWAIT);
            else if (hidx == OP_SCALE) { if (code_at(pc) != OP_SCALE) bad = plk_fmt("pt program: word %ld is not the program's op", (long)wi); pc++; }
            else {
                if (hidx >= 24) { bad = product(wi); if (!bad.empty()) return bad; }
                const bool push = hidx >= 24 ? hidx < 28 : hidx < 16;
                if (code_at(pc) != (push ? OP_PUSH : OP_POPMUL) || (int)y >= 4 || (int)y != pg.ops[pc].y || (full[y] != 0) == push)
                    bad = plk_fmt(push ? "pt program: bad PUSH at word %ld" : "pt program: bad POPMUL at word %ld", (long)wi);
                else full[y] = push;
                pc++;
            }
            if (!bad.empty()) return bad;
        }
        waited = true;
    }
}

/*
 * 64-bit ops of the two-sites-per-lane interpreter k_ll_fused4_v4 (plk_fused4_v4.h), a re-encoding of PlkFusedPT::words:
 *   lo = handler index * 512;  hi (observation ops) = (LDS offset / 32 of the next observation's table) << 16 |
 *        (offset / 64 of the code row after next, relative to row 0);  tip_base = LDS byte address of the table image.
 */
#define PLK_V4_HANDLER_BYTES 512
#define PLK_V4_HANDLER_SLOTS 64
#define PLK_V4_REFILL_A 32
#define PLK_V4_REFILL_B 33
struct PlkFusedV4 {
    std::vector<unsigned> words;       /* 2 dwords per op, blocks of 7 ops + 1 REFILL op, two spare blocks */
    unsigned first_y = 0, first_z = 0, second_z = 0;
};

static inline bool plk_word_is_obs(unsigned hidx)
{
    return hidx == OP_TIP_SET || hidx == OP_TIP_MUL || hidx == PLK_WORD_TIPMUL_NOWAIT || hidx == PLK_WORD_MATVEC_TIPMUL ||
           (hidx >= PLK_WORD_SET_POPMUL && hidx < PLK_WORD_SET_POPMUL + 4) || (hidx >= PLK_WORD_SET_PUSH && hidx < PLK_WORD_SET_PUSH + 4);
}

/* Threaded op stream of k_ll_fused4_v4: blocks of 8 ops = 7 ops of the program + one REFILL op (REFILL_A in even blocks,
 * REFILL_B in odd ones: the handler that requests the block after next into its own half of the kernel's op ring); the
 * program's ops up to its END, END padding to the end of that block, then two spare blocks (the ring reads two ahead). */
static inline void plk_fused_v4_words(const PlkFusedPT &fu, int nchar, int tile, unsigned tip_base, PlkFusedV4 &v4)
{
    const unsigned tb32 = tip_base / 32, rg = (unsigned)tile / 64;
    size_t nreal = 0;
    while (nreal < fu.words.size() && (fu.words[nreal] & 31) != OP_END) nreal++;
    nreal++;                                                  /* the END itself */
    const size_t nblocks = (nreal + 6) / 7 + 2;
    v4.words.assign(nblocks * 16, 0u);
    size_t src = 0;
    for (size_t b = 0; b < nblocks; b++)
        for (int j = 0; j < 8; j++) {
            unsigned *o = &v4.words[(b * 8 + j) * 2];
            if (j == 7) { o[0] = (unsigned)((b & 1) ? PLK_V4_REFILL_B : PLK_V4_REFILL_A) * PLK_V4_HANDLER_BYTES; continue; }
            const unsigned w = src < nreal && src < fu.words.size() ? fu.words[src] : (unsigned)OP_END, hidx = w & 31;
            src++;
            o[0] = hidx * PLK_V4_HANDLER_BYTES;
            if (plk_word_is_obs(hidx)) o[1] = ((tb32 + ((w >> 5) & 0x7ff) * (unsigned)nchar) << 16) | ((w >> 16) * rg);
        }
    v4.first_y = tb32 + (unsigned)fu.first_unit * (unsigned)nchar;
    v4.first_z = (unsigned)fu.first_row * rg;
    v4.second_z = (unsigned)fu.second_row * rg;
}

/* the 32-bit program is checked by plk_fused_check_pt; this checks the re-encoding: the stream is the program's ops in
 * order with a REFILL op of the right kind closing every block, an END is reached, two whole blocks follow the block of the
 * END (what the ring has requested by then), every field fits its 16 bits, is the 32-bit word's field times its granule,
 * and the byte addresses the kernel forms from it stay inside the launch's LDS */
static inline std::string plk_fused_check_v4(const PlkFusedPT &fu, const PlkFusedV4 &v4, int nchar, int tile, unsigned tip_base,
                                             size_t lds_bytes_launched)
{
    if (tip_base % 32 != 0 || tile % 64 != 0) return "v4 program: LDS base or tile granule";
    if (v4.words.size() % 16 != 0 || v4.words.size() < 48) return "v4 program: word count";
    const size_t tip_bytes = (size_t)fu.units * nchar * 32, lds_end = tip_base + lds_bytes_launched;
    const unsigned tb32 = tip_base / 32, rg = (unsigned)tile / 64;
    auto y_ok = [&](unsigned y, unsigned unit) { return y == tb32 + unit * (unsigned)nchar && y < 65536u && (size_t)y * 32 >= tip_base && (size_t)y * 32 < tip_base + tip_bytes; };
    auto z_ok = [&](unsigned z, unsigned row) { return z == row * rg && z < 65536u && tip_base + tip_bytes + (size_t)z * 64 + (size_t)tile <= lds_end; };
    if (!y_ok(v4.first_y, (unsigned)fu.first_unit) || !z_ok(v4.first_z, (unsigned)fu.first_row) || !z_ok(v4.second_z, (unsigned)fu.second_row)) return "v4 program: prologue fields";
    const size_t nblocks = v4.words.size() / 16;
    size_t src = 0;
    long end_block = -1;
    for (size_t b = 0; b < nblocks; b++)
        for (int j = 0; j < 8; j++) {
            const unsigned lo = v4.words[(b * 8 + j) * 2], hi = v4.words[(b * 8 + j) * 2 + 1];
            if (lo % PLK_V4_HANDLER_BYTES != 0 || lo >= (unsigned)PLK_V4_HANDLER_SLOTS * PLK_V4_HANDLER_BYTES) return plk_fmt("v4 program: handler offset in block %ld", (long)b);
            const unsigned hidx = lo / PLK_V4_HANDLER_BYTES;
            if (j == 7) {
                if (hidx != (unsigned)((b & 1) ? PLK_V4_REFILL_B : PLK_V4_REFILL_A) || hi != 0) return plk_fmt("v4 program: block %ld does not end in its REFILL op", (long)b);
                continue;
            }
            if (hidx >= 32) return plk_fmt("v4 program: REFILL op inside block %ld", (long)b);
            if (end_block >= 0) { if (hidx != OP_END || hi != 0) return "v4 program: ops after END"; continue; }
            if (src >= fu.words.size()) return "v4 program: more ops than the program";
            const unsigned w = fu.words[src++];
            if (hidx != (w & 31)) return plk_fmt("v4 program: op %ld is not the program's", (long)(src - 1));
            if (hidx == OP_END) { if (hi != 0) return "v4 program: stray fields in END"; end_block = (long)b; continue; }
            if (!plk_word_is_obs(hidx)) { if (hi != 0) return plk_fmt("v4 program: stray fields in op %ld", (long)(src - 1)); continue; }
            if (!y_ok(hi >> 16, (w >> 5) & 0x7ff) || !z_ok(hi & 0xffff, w >> 16)) return plk_fmt("v4 program: fields of op %ld", (long)(src - 1));
        }
    if (end_block < 0) return "v4 program: no END";
    if ((size_t)end_block + 2 >= nblocks) return "v4 program: fewer than two spare blocks after the END";
    return "";
}

/*
 * The vector ll kernel (k_ll_vec, plk_vec.h) on the pair-table program: int4 ops
 *   observation (TIP_SET / TIP_MUL)  x = opcode, y = first row of the op's table (unit * nchar: the kernel adds the code),
 *                                    w = staged row of the NEXT observation op (cyclic)
 *   MATVEC                           x = opcode, z = op index of the next MATVEC (cyclic; its matrix lines are touched ahead)
 *   PUSH / POPMUL y = slot;  SCALE
 * op_edge[pc] = CSR edge of a MATVEC op (-1 otherwise): the matrix stream is indexed by op.
 */
struct PlkVecPT {
    std::vector<plk_op4> ops;
    std::vector<int> op_edge;
    int first_base = 0, first_row = 0;
};

static inline void plk_vec_pt_build(const PlkFusedPT &fu, int nchar, PlkVecPT &vp)
{
    const int n = (int)fu.vops.size();
    vp.ops.assign(n, plk_op4{OP_END, 0, 0, 0});
    vp.op_edge.assign(n, -1);
    std::vector<int> obs_idx, mv_idx;
    for (int i = 0; i < n; i++) {
        const PlkFusedPT::VOp &o = fu.vops[i];
        vp.ops[i].x = o.code;
        if (o.code == OP_TIP_SET || o.code == OP_TIP_MUL) { vp.ops[i].y = o.unit * nchar; obs_idx.push_back(i); }
        else if (o.code == OP_MATVEC) { vp.op_edge[i] = o.edge; mv_idx.push_back(i); }
        else if (o.code == OP_PUSH || o.code == OP_POPMUL) vp.ops[i].y = o.d;
    }
    for (size_t q = 0; q < obs_idx.size(); q++) vp.ops[obs_idx[q]].w = fu.vops[obs_idx[(q + 1) % obs_idx.size()]].row;
    for (size_t q = 0; q < mv_idx.size(); q++) vp.ops[mv_idx[q]].z = mv_idx[(q + 1) % mv_idx.size()];
    vp.first_base = obs_idx.empty() ? 0 : fu.vops[obs_idx[0]].unit * nchar;
    vp.first_row = obs_idx.empty() ? 0 : fu.vops[obs_idx[0]].row;
}

/* replay against the program: the ops fold exactly the program's ops (a pair look-up stands for a cherry's three ops), rows
 * and tables are the ones of the op, every index is in range */
static inline std::string plk_vec_pt_check(int N, const int *ip, const int *ix, const PlkProgram &pg, const PlkFusedPT &fu, const PlkVecPT &vp,
                                           int nchar, int E)
{
    const int nops = (int)pg.ops.size(), n = (int)vp.ops.size(), nrows = (int)fu.row_node.size(), ntab = (int)fu.tab_unit.size();
    if ((int)vp.op_edge.size() != n || (int)fu.vops.size() != n) return "vec pt program: sizes";
    std::vector<int> table_at(std::max(fu.units, 1), -1);
    for (int t = 0; t < ntab; t++) if (fu.tab_unit[t] >= 0 && fu.tab_unit[t] < fu.units) table_at[fu.tab_unit[t]] = t;
    size_t pc = 0;
    long next_row = vp.first_row, next_base = vp.first_base;
    bool first = true;
    std::vector<char> full(std::max(pg.slots_needed, 1), 0);
    int last_mv = -1, first_mv = -1;
    for (int i = 0; i < n; i++) {
        const plk_op4 &o = vp.ops[i];
        const int code = o.x & 0xff;
        if (pc >= (size_t)nops) return "vec pt program: more ops than the program";
        const int pcode = pg.ops[pc].x & 0xff;
        if (code == OP_TIP_SET || code == OP_TIP_MUL) {
            if (o.y % nchar != 0 || o.y / nchar < 0 || o.y / nchar >= fu.units || table_at[o.y / nchar] < 0) return plk_fmt("vec pt program: op %ld names no table", i);
            const int t = table_at[o.y / nchar];
            if (o.w < 0 || o.w >= nrows) return plk_fmt("vec pt program: op %ld prefetches a row out of range", i);
            if (next_row < 0 || next_row >= nrows || (!first && false)) return "vec pt program: row chain";
            if (next_base != o.y) return plk_fmt("vec pt program: op %ld: the chain's table is not the op's", i);
            const int row = (int)next_row;
            if (fu.tab_eb[t] >= 0) {
                if (code != OP_TIP_SET || !plk_pair_at(N, ip, ix, pg, pc)) return plk_fmt("vec pt program: op %ld is a pair look-up where the program has no cherry", i);
                if (fu.tab_eb[t] != pg.op_edge[pc] || fu.tab_ec[t] != pg.op_edge[pc + 1] || fu.tab_edge[t] != pg.op_edge[pc + 2]) return plk_fmt("vec pt program: pair table of op %ld", i);
                if (fu.row_node[row] != pg.ops[pc].y || fu.row_node2[row] != pg.ops[pc + 1].y) return plk_fmt("vec pt program: pair row of op %ld", i);
                pc += 3;
            } else {
                if (pcode != OP_TIP_SET && pcode != OP_TIP_MUL && pcode != OP_NODE_MUL) return plk_fmt("vec pt program: op %ld is not an observation", i);
                if ((code == OP_TIP_SET) != (pcode == OP_TIP_SET)) return plk_fmt("vec pt program: op %ld SET/MUL", i);
                if (fu.row_node[row] != pg.ops[pc].y || fu.row_node2[row] != -1) return plk_fmt("vec pt program: row of op %ld", i);
                if (fu.tab_edge[t] != (pcode == OP_NODE_MUL ? -1 : pg.op_edge[pc])) return plk_fmt("vec pt program: table of op %ld", i);
                pc++;
            }
            /* the chain: this op names the row of the next observation; its table base is found at that op */
            next_row = o.w;
            next_base = -1;
            for (int q = 1; q <= n; q++) {
                const plk_op4 &o2 = vp.ops[(i + q) % n];
                if ((o2.x & 0xff) == OP_TIP_SET || (o2.x & 0xff) == OP_TIP_MUL) { next_base = o2.y; break; }
            }
            first = false;
        } else if (code == OP_MATVEC) {
            if (pcode != OP_MATVEC || vp.op_edge[i] != pg.op_edge[pc] || vp.op_edge[i] < 0 || vp.op_edge[i] >= E) return plk_fmt("vec pt program: product %ld", i);
            if (o.z < 0 || o.z >= n || (vp.ops[o.z].x & 0xff) != OP_MATVEC) return plk_fmt("vec pt program: op %ld names no next product", i);
            if (first_mv < 0) first_mv = i;
            last_mv = i;
            pc++;
        } else if (code == OP_PUSH || code == OP_POPMUL) {
            if (pcode != code || o.y != pg.ops[pc].y || o.y < 0 || o.y >= (int)full.size() || (full[o.y] != 0) == (code == OP_PUSH)) return plk_fmt("vec pt program: stack op %ld", i);
            full[o.y] = code == OP_PUSH;
            pc++;
        } else if (code == OP_SCALE) {
            if (pcode != OP_SCALE) return plk_fmt("vec pt program: op %ld", i);
            pc++;
        } else return plk_fmt("vec pt program: unknown op %ld", i);
    }
    if ((int)pc != nops) return "vec pt program: the program's ops are not all covered";
    if (last_mv >= 0 && vp.ops[last_mv].z != first_mv) return "vec pt program: the product chain does not wrap to the first";
    return "";
}

/* the C++ interpreter k_ll_fused4<D, NS> (plk_fused4.h): pair-ahead fetch, one-ahead code row chain */
static inline std::string plk_fused_check_cpp(int N, const PlkProgram &pg, const PlkFused &fu, int nchar, int D, int NS,
                                              size_t lds_bytes_launched)
{
    const int nops = (int)pg.ops.size(), ntips1 = (int)pg.tip_edge.size() + 1, nobs = (int)pg.obs_nodes.size();
    if (D != 4 && D != 8 && D != 16) return "fused program: stack depth variant";
    if (pg.slots_needed > D) return "fused program: the tree needs a deeper register stack";
    if (nchar < 1 || nchar > 256) return "fused program: nchar out of range";
    if ((size_t)ntips1 * nchar * 32 + (size_t)nobs * PLK_TILE * NS > lds_bytes_launched) return "fused program: LDS image larger than the launch's dynamic LDS";
    if (lds_bytes_launched > PLK_LDS_LIMIT) return "fused program: dynamic LDS above the limit";
    /* the loop runs over even pc < nops and fetches the pair (pc + 2, pc + 3) while it executes (pc, pc + 1) */
    const int last_pc = nops > 0 ? ((nops - 1) / 2) * 2 : 0;
    if (fu.fops.size() < 2 || (nops > 0 && (size_t)(last_pc + 3) >= fu.fops.size())) return "fused program: op buffer too short for the look-ahead";
    std::vector<int> slot, rowv, opc;
    plk_obs_sequence(N, pg, slot, rowv, opc);
    if (fu.first_row < 0 || fu.first_row >= std::max(nobs, 1)) return "fused program: first row out of range";
    long cur_row = fu.first_row;
    size_t oi = 0;
    int mi = 0;
    for (int pc = 0; pc < nops; pc++) {
        const plk_op4 f = fu.fops[pc];
        const int code = f.x & 0xff;
        if (code != (pg.ops[pc].x & 0xff)) return "fused program: opcode mismatch";
        if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
            if (oi >= slot.size() || cur_row != rowv[oi]) return plk_fmt("fused program: code chain delivers the wrong row at op %ld", pc);
            if (code != OP_NODE_MUL && ((f.x >> 8) < 0 || (f.x >> 8) >= ntips1 - 1)) return "fused program: tip slot out of range";
            if (f.z < 0 || f.z >= nobs) return "fused program: next row out of range";
            cur_row = f.z;
            oi++;
        } else if (code == OP_MATVEC) {
            if (f.w != mi || fu.mat_edge[mi] != pg.op_edge[pc]) return "fused program: matrix index mismatch";
            mi++;
        } else if (code == OP_PUSH || code == OP_POPMUL) {
            if (f.y < 0 || f.y >= D) return "fused program: stack slot out of range";
        }
    }
    for (size_t pc = nops; pc < fu.fops.size(); pc++) if ((fu.fops[pc].x & 0xff) != OP_END) return "fused program: padding is not END";
    return "";
}

/* ------------------------------------------------------------------------------------------------------------ */
/* down passes that traverse the program (k_down_fused4, k_ll_mfma, k_down_fused_mfma): int4 ops with an          */
/* observation chain (z = tip slot of the next observation op, w = its staged row; the last one wraps with bit 30) */
/* ------------------------------------------------------------------------------------------------------------ */

struct PlkChain {
    std::vector<plk_op4> ops;          /* + one trailing OP_END (k_down_fused4 reads one op ahead) */
    int first_slot = -1, first_row = 0;
    int second_row = 0;                /* mode 3: staged row of the second observation op */
};

/* mode 3 (k_ll_vec, k_down_vec): as mode 1, and every observation op also carries in y the staged row of the observation
 * op AFTER the next one (the prefetch chain runs two ops ahead: value of the next op, code of the one after; the
 * sequence wraps to the start for the next category); MATVEC w = CSR edge of the next MATVEC.
 * mode 0: MATVEC keeps (x, y) of the program, z = op index of the next MATVEC, wrapping (k_ll_mfma); mode 1: MATVEC y = CSR edge, z = storage index of the
 * child node, w = CSR edge of the next MATVEC, wrapping to the first (k_down_fused4, k_down_vec); mode 2: as 1 but
 * w = storage index of the edge (k_down_fused_mfma, no chain).  SCALE y = rescaling slot of the node or -1 (modes 1, 2). */
static inline void plk_chain_build(int N, const PlkProgram &pg, int mode, const int *indices, const int *node_int,
                                   const int *edge_int, const int *node_scale, PlkChain &ch, const char *skip_store = nullptr)
{
    const int ntips = (int)pg.tip_edge.size();
    std::vector<int> row(N, -1);
    for (size_t r = 0; r < pg.obs_nodes.size(); r++) row[pg.obs_nodes[r]] = (int)r;
    ch.ops.assign(pg.ops.size() + 1, plk_op4{OP_END, 0, 0, 0});
    ch.first_slot = -1; ch.first_row = 0;
    int prev = -1;
    for (size_t pc = 0; pc < pg.ops.size(); pc++) {
        const int code = pg.ops[pc].x & 0xff;
        plk_op4 o = {pg.ops[pc].x, pg.ops[pc].y, 0, 0};
        if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
            const int slot = code == OP_NODE_MUL ? ntips : (o.x >> 8);
            o.y = row[pg.ops[pc].y];
            if (mode != 2) {
                if (prev < 0) { ch.first_slot = slot; ch.first_row = o.y; }
                else { ch.ops[prev].z = slot; ch.ops[prev].w = o.y; }
                prev = (int)pc;
            }
        } else if (code == OP_MATVEC && mode >= 1) {
            o.y = pg.op_edge[pc];
            o.z = node_int ? node_int[indices[o.y]] : 0;
            if (mode >= 1 && skip_store && skip_store[indices[o.y]]) o.z = -1;      /* this child's vector is not stored */
            if (mode == 2) o.w = edge_int[o.y];
        } else if (code == OP_SCALE && mode >= 1 && node_scale) {
            o.y = node_scale[pg.ops[pc].y];
        }
        ch.ops[pc] = o;
    }
    if (prev >= 0) { ch.ops[prev].z = ch.first_slot | (1 << 30); ch.ops[prev].w = ch.first_row; }
    if (mode == 3) {
        /* y of an observation op = staged row of the op two observations later (cyclically) */
        std::vector<int> opc_, rows_;
        for (size_t pc = 0; pc < pg.ops.size(); pc++) {
            const int code = pg.ops[pc].x & 0xff;
            if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) { opc_.push_back((int)pc); rows_.push_back(ch.ops[pc].y); }
        }
        const size_t no = opc_.size();
        ch.second_row = no ? rows_[1 % no] : 0;
        for (size_t i = 0; i < no; i++) ch.ops[opc_[i]].y = rows_[(i + 2) % no];
    }
    if (mode == 0) {
        /* MATVEC z = op index of the next MATVEC, wrapping to the first (k_ll_mfma requests its matrix fragments early) */
        int first_mv = -1, prev_mv = -1;
        for (size_t pc = 0; pc < pg.ops.size(); pc++)
            if ((pg.ops[pc].x & 0xff) == OP_MATVEC) {
                if (first_mv < 0) first_mv = (int)pc;
                if (prev_mv >= 0) ch.ops[prev_mv].z = (int)pc;
                prev_mv = (int)pc;
            }
        if (prev_mv >= 0) ch.ops[prev_mv].z = first_mv;
    }
    if (mode == 1 || mode == 3) {
        /* MATVEC w = CSR edge of the next MATVEC (wrapping); without node storage (ll kernels) z = its op index */
        int first_mv = -1, prev_mv = -1;
        for (size_t pc = 0; pc < pg.ops.size(); pc++)
            if ((pg.ops[pc].x & 0xff) == OP_MATVEC) {
                if (first_mv < 0) first_mv = (int)pc;
                if (prev_mv >= 0) { ch.ops[prev_mv].w = pg.op_edge[pc]; if (!node_int) ch.ops[prev_mv].z = (int)pc; }
                prev_mv = (int)pc;
            }
        if (prev_mv >= 0) { ch.ops[prev_mv].w = pg.op_edge[first_mv]; if (!node_int) ch.ops[prev_mv].z = first_mv; }
    }
}

/*
 * Up pass of the vector kernels (k_up_vec, plk_updown_vec.h): visit records of the internal nodes in BFS order and
 * the list of matrices in the exact order the kernel consumes them (kind 0: P for a child message, 1: the edge-form
 * matrix, 2: P for the forward vector of a child).  The builder and the kernel are the two halves of one contract;
 * plk_up_visits_check() replays the kernel's consumption against the list.
 *   header (8 ints): node, number of children, storage index of the node, rescaling slot or -1, has_data,
 *                    first CSR edge, marginal of the root wanted (first record only), 0
 *   child (4 ints):  node, tip slot or -1, PLK_UP_* flags, storage index (internal child) or -1
 * An internal child whose own children (one or two) are all leaves is handled INSIDE this visit when no marginals are
 * asked for (PLK_UP_INLINE): its forward vector is used while it is in registers and never stored, the node gets no
 * visit of its own, and the child's fourth int is the offset (in rec) of a 12-int record placed after all visits:
 *   node, has_data, rescaling slot or -1, number of leaves (1 or 2), first CSR edge,
 *   leaf 0: node, tip slot, wanted (bit 0 derivative, bit 1 marginal);  leaf 1: node, tip slot, wanted;  0
 * With marginals a child is inlined only if all its leaves observe single states at every site (leaf_observed).
 */
enum { PLK_UP_WANT_D = 1, PLK_UP_WANT_F = 2, PLK_UP_WANT_M = 4, PLK_UP_STORE_F = 8, PLK_UP_INLINE = 16 };

/* a node whose one or two children are all leaves is finished inside its parent's visit when no marginals are asked for:
 * its forward vector is never stored, and neither is its L vector (the up pass rebuilds it from the tip tables) */
static inline bool plk_up_inlinable(const int *ip, const int *edge_tip, int b, bool marg, const int *ix = nullptr,
                                    const char *leaf_observed = nullptr)
{
    const int d = ip[b + 1] - ip[b];
    if (d < 1 || d > 2) return false;
    /* with marginals: only when every site of every leaf below observes a single state (leaf_observed, from the pattern
     * upload) -- the leaf's marginal is then one dot product per site, done in the parent's parent's visit too */
    if (marg && (!ix || !leaf_observed)) return false;
    for (int idx = ip[b]; idx < ip[b + 1]; idx++) {
        if (edge_tip[idx] < 0) return false;
        if (marg && !leaf_observed[ix[idx]]) return false;
    }
    return true;
}

struct PlkUpVisits {
    std::vector<int> rec;              /* visit records, then the inline records */
    size_t visit_ints = 0;             /* ints of rec taken by the visit records */
    int nvisits = 0;
    std::vector<int> kind, edge;       /* the matrix stream */
};

static inline void plk_up_visits_build(int N, const int *ip, const int *ix, const int *preorder, const char *node_has_data,
                                       const int *edge_tip, const int *node_int, const int *node_scale, bool deriv, bool marg,
                                       const int *edge_mask, const int *node_mask, PlkUpVisits &uv, const char *leaf_observed = nullptr)
{
    uv.rec.clear(); uv.kind.clear(); uv.edge.clear(); uv.nvisits = 0;
    auto flags = [&](int idx, int b) {
        const bool leaf = edge_tip[idx] >= 0;
        const bool wd = deriv && (!edge_mask || edge_mask[idx]);
        const bool wm = marg && (!node_mask || node_mask[b]);
        const bool wf = !leaf || wm;
        return (wd ? PLK_UP_WANT_D : 0) | (wf ? PLK_UP_WANT_F : 0) | (wm ? PLK_UP_WANT_M : 0) | (!leaf ? PLK_UP_STORE_F : 0);
    };
    auto put = [&](int kind, int edge) { uv.kind.push_back(kind); uv.edge.push_back(edge); };
    auto inlinable = [&](int b) { return plk_up_inlinable(ip, edge_tip, b, marg, ix, leaf_observed); };
    std::vector<int> inl;                /* inline records, appended after the visits */
    std::vector<size_t> fix;             /* positions in rec that hold an offset into inl */
    for (int u = 0; u < N; u++) {
        const int a = preorder[u];
        const int start = ip[a], deg = ip[a + 1] - start;
        if (deg == 0) continue;
        if (u > 0 && inlinable(a)) continue;             /* handled inside its parent's visit */
        const int root_m = u == 0 && marg && (!node_mask || node_mask[a]);
        const int hdr[8] = {a, deg, node_int[a], node_scale[a], node_has_data[a] ? 1 : 0, start, root_m, 0};
        uv.rec.insert(uv.rec.end(), hdr, hdr + 8);
        for (int j = 0; j < deg; j++) {
            const int idx = start + j, b = ix[idx];
            int fl = flags(idx, b), second = edge_tip[idx];
            const int fourth = edge_tip[idx] >= 0 ? -1 : node_int[b];
            if (edge_tip[idx] < 0 && inlinable(b)) {
                fl = (fl & ~PLK_UP_STORE_F) | PLK_UP_INLINE | PLK_UP_WANT_F;
                const int s0 = ip[b], db = ip[b + 1] - s0;
                int r[12] = {b, node_has_data[b] ? 1 : 0, node_scale[b], db, s0, 0, 0, 0, 0, 0, 0, 0};
                for (int q = 0; q < db; q++) {
                    r[5 + 3 * q] = ix[s0 + q];
                    r[6 + 3 * q] = edge_tip[s0 + q];
                    r[7 + 3 * q] = (deriv && (!edge_mask || edge_mask[s0 + q]) ? 1 : 0) | (marg && (!node_mask || node_mask[ix[s0 + q]]) ? 2 : 0);
                }
                second = (int)inl.size();            /* becomes -2 - (visit_ints + offset) below */
                fix.push_back(uv.rec.size() + 1);
                inl.insert(inl.end(), r, r + 12);
            }
            const int cr[4] = {b, second, fl, fourth};
            uv.rec.insert(uv.rec.end(), cr, cr + 4);
        }
        uv.nvisits++;
        for (int j = 0; j < deg; j++) {
            const int idx = start + j, fl = flags(idx, ix[idx]) | (edge_tip[idx] < 0 && inlinable(ix[idx]) ? PLK_UP_WANT_F : 0);
            if (!(fl & (PLK_UP_WANT_D | PLK_UP_WANT_F))) continue;
            for (int j2 = 0; j2 < deg; j2++)
                if (j2 != j && edge_tip[start + j2] < 0) put(0, start + j2);
            if ((fl & PLK_UP_WANT_D) && edge_tip[idx] < 0) put(1, idx);
            if (fl & PLK_UP_WANT_F) put(2, idx);
        }
    }
    uv.visit_ints = uv.rec.size();
    for (size_t f : fix) uv.rec[f] = -2 - (uv.rec[f] + (int)uv.visit_ints);
    uv.rec.insert(uv.rec.end(), inl.begin(), inl.end());
}

/* replays k_up_vec's walk over the records: every index in range, every stored forward vector written before it is
 * read, and the matrices consumed are exactly the list, in order */
static inline std::string plk_up_visits_check(int N, int E, const PlkUpVisits &uv, int nint_nodes, int ntips, int nscale_slots,
                                              bool deriv)
{
    std::vector<char> f_written(std::max(nint_nodes, 1), 0), visited(N, 0), inlined(N, 0);
    size_t vp = 0, ms = 0;
    for (int v = 0; v < uv.nvisits; v++) {
        if (vp + 8 > uv.rec.size()) return "up visits: record overrun";
        const int *h = &uv.rec[vp];
        const int a = h[0], deg = h[1], ai = h[2], slot = h[3], e0 = h[5];
        if (a < 0 || a >= N || deg < 1 || ai < 0 || ai >= nint_nodes || slot < -1 || slot >= nscale_slots) return plk_fmt("up visits: bad header %ld", v);
        if (e0 < 0 || e0 + deg > E) return plk_fmt("up visits: bad edge range in visit %ld", v);
        if (vp + 8 + 4 * (size_t)deg > uv.rec.size()) return "up visits: record overrun";
        if (v == 0) f_written[ai] = 1;       /* the root's forward vector is written first */
        if (!f_written[ai]) return plk_fmt("up visits: forward vector of node %ld read before it is written", a);
        visited[a] = 1;
        const int *ch = h + 8;
        auto need = [&](int kind, int edge) -> bool { const bool ok = ms < uv.kind.size() && uv.kind[ms] == kind && uv.edge[ms] == edge; ms++; return ok; };
        for (int j = 0; j < deg; j++) {
            const int b = ch[4 * j], t = ch[4 * j + 1], fl = ch[4 * j + 2], bi = ch[4 * j + 3];
            if (b < 0 || b >= N || t >= ntips || ((t < -1) != ((fl & PLK_UP_INLINE) != 0))) return plk_fmt("up visits: bad child in visit %ld", v);
            if (t < 0 && (bi < 0 || bi >= nint_nodes)) return plk_fmt("up visits: bad storage index in visit %ld", v);
            if (fl & PLK_UP_INLINE) {
                /* the child's leaves are handled here: its record must lie behind the visits and be in range */
                const long off = -2 - (long)t;
                if ((fl & PLK_UP_STORE_F) || !(fl & PLK_UP_WANT_F)) return plk_fmt("up visits: flags of an inline child in visit %ld", v);
                if (off < (long)uv.visit_ints || (size_t)off + 12 > uv.rec.size()) return plk_fmt("up visits: inline record out of range in visit %ld", v);
                const int *q = &uv.rec[off];
                if (q[0] != b || q[2] < -1 || q[2] >= nscale_slots || q[3] < 1 || q[3] > 2 || q[4] < 0 || q[4] + q[3] > E) return plk_fmt("up visits: bad inline record in visit %ld", v);
                for (int l = 0; l < q[3]; l++)
                    if (q[5 + 3 * l] < 0 || q[5 + 3 * l] >= N || q[6 + 3 * l] < 0 || q[6 + 3 * l] >= ntips) return plk_fmt("up visits: bad inline leaf in visit %ld", v);
                inlined[b] = 1;
                continue;
            }
            if (((fl & PLK_UP_STORE_F) != 0) != (t < 0) || (t < 0 && !(fl & PLK_UP_WANT_F))) return plk_fmt("up visits: flags of an internal child in visit %ld", v);
            if (!deriv && (fl & PLK_UP_WANT_D)) return "up visits: derivative flag without a derivative pass";
        }
        for (int j = 0; j < deg; j++) {
            const int fl = ch[4 * j + 2];
            if (!(fl & (PLK_UP_WANT_D | PLK_UP_WANT_F))) continue;
            for (int j2 = 0; j2 < deg; j2++)
                if (j2 != j && ch[4 * j2 + 1] < 0 && !need(0, e0 + j2)) return plk_fmt("up visits: stream mismatch (sibling) in visit %ld", v);
            if ((fl & PLK_UP_WANT_D) && ch[4 * j + 1] < 0 && !need(1, e0 + j)) return plk_fmt("up visits: stream mismatch (edge form) in visit %ld", v);
            if ((fl & PLK_UP_WANT_F) && !need(2, e0 + j)) return plk_fmt("up visits: stream mismatch (forward) in visit %ld", v);
        }
        for (int j = 0; j < deg; j++) if (ch[4 * j + 2] & PLK_UP_STORE_F) f_written[ch[4 * j + 3]] = 1;
        vp += 8 + 4 * (size_t)deg;
    }
    if (vp != uv.visit_ints) return "up visits: trailing records";
    for (int a = 0; a < N; a++) if (visited[a] && inlined[a]) return plk_fmt("up visits: node %ld is both visited and inlined", a);
    if (ms != uv.kind.size()) return "up visits: matrix stream not consumed";
    for (size_t i = 0; i < uv.edge.size(); i++) if (uv.edge[i] < 0 || uv.edge[i] >= E || uv.kind[i] < 0 || uv.kind[i] > 2) return "up visits: bad stream entry";
    return "";
}

/*
 * Node visits of k_up_nodes_mfma (plk_mfma_updown.h): the derivative pass without marginals.  One visit per internal
 * node a; what is stored per internal node is the vector G_a at the TOP of the edge into a (forward vector of the
 * parent with the parent's observation, rescaling factor and the sibling messages folded in), not the forward vector
 * at a.  A visit finishes the derivative of a's own edge from G_a and the child messages (stored edge vectors / tip
 * tables; L_a is not read), forms F_a = P_a^T G_a and the G_b of every child.  Visits are in depth-first order; the G
 * of the last internal child that still has work below it is not stored but handed to the next visit in registers
 * (PLK_UN_CONTINUE on the child -- always the LAST child record -- and PLK_UN_FROM_REGS on the next header).
 *   header (8 ints): node, number of children, storage index, rescaling slot or -1, has_data, first CSR edge,
 *                    CSR edge into the node (-1: root, whose visit is the first and takes the root prior from
 *                    registers), PLK_UN_* header flags
 *   child (4 ints):  node, tip slot or -1, PLK_UN_* child flags | position among the node's CSR children << 4,
 *                    storage index (internal child) or -1
 */
enum { PLK_UN_OWN_D = 1, PLK_UN_FROM_REGS = 2 };                                              /* header flags */
enum { PLK_UN_LEAF_D = 1, PLK_UN_STORE_G = 2, PLK_UN_CONTINUE = 4, PLK_UN_PAIR = 8, PLK_UN_REBUILD = 16,
       PLK_UN_INL_OWN = 32, PLK_UN_INL_L0 = 64, PLK_UN_INL_L1 = 128, PLK_UN_INL = 224, PLK_UN_WORK = 7 | 224, PLK_UN_POS_SHIFT = 8 };   /* child flags;
   PLK_UN_INL_*: a pair child that is FINISHED INSIDE this visit (it gets no visit of its own and its G is neither stored nor
   handed on): the derivative of the edge into it (OWN) and of its first / second leaf edge (L0 / L1) are wanted;
   PLK_UN_PAIR: the child's message P_b L_b comes from pair table number `fifth field` (its two leaves' codes), L_b is not stored;
   the record's fourth int then still holds the storage index (the child's own visit needs its G);
   PLK_UN_REBUILD: L_b is not stored either: b has two children, each a leaf or a pair node, and L_b is the product of their
   two messages (tip-table row / pair-table row), listed in the rebuild table at 4 * storage index (plk_up_rebuild_table) */

struct PlkUpNodes {
    std::vector<int> rec;
    int nvisits = 0;
};

/* Nodes whose L vector the k = 4 node-visit passes rebuild from tables instead of moving it through HBM: two children,
 * each a leaf or a pair node, no data of its own, no rescaling at the node (with <= 6 edges below it, it never is one).
 * rebuild[b] = 1 for them; table[4 * node_int[b]] = {t0, n0, t1, n1}: t >= 0 the tip slot of leaf child n, t <= -2 the pair
 * table -2 - t of pair child n. */
static inline void plk_up_rebuild_table(int N, const int *ip, const int *ix, const char *node_has_data, const int *edge_tip,
                                        const int *node_int, const int *node_scale, const int *pair_of, int nint_nodes,
                                        std::vector<char> &rebuild, std::vector<int> &table)
{
    rebuild.assign(N, 0);
    table.assign(4 * (size_t)std::max(nint_nodes, 1), -1);
    std::vector<int> edge_into(N, -1);
    for (int a = 0; a < N; a++) for (int idx = ip[a]; idx < ip[a + 1]; idx++) edge_into[ix[idx]] = idx;
    for (int b = 0; b < N; b++) {
        const int e0 = ip[b];
        if (ip[b + 1] - e0 != 2 || edge_into[b] < 0 || node_has_data[b] || node_scale[b] >= 0 || pair_of[b] >= 0 || node_int[b] < 0) continue;
        int rec[4];
        bool ok = true;
        for (int j = 0; j < 2; j++) {
            const int ch = ix[e0 + j];
            if (edge_tip[e0 + j] >= 0) rec[2 * j] = edge_tip[e0 + j];
            else if (pair_of[ch] >= 0) rec[2 * j] = -2 - pair_of[ch];
            else ok = false;
            rec[2 * j + 1] = ch;
        }
        if (!ok) continue;
        rebuild[b] = 1;
        for (int q = 0; q < 4; q++) table[4 * (size_t)node_int[b] + q] = rec[q];
    }
}

/* pair_of: null, or N ints: index of the node's pair table (its message comes from the table, not from a stored L) or -1.
 * A pair child's flags carry PLK_UN_PAIR and its tip-slot field holds -2 - pair (tip slots are >= 0, -1 = internal).
 * rebuild: null, or N chars of plk_up_rebuild_table: such a child's flags carry PLK_UN_REBUILD. */
static inline void plk_up_nodes_build(int N, const int *ip, const int *ix, const int *preorder, const char *node_has_data,
                                      const int *edge_tip, const int *node_int, const int *node_scale, const int *edge_mask,
                                      PlkUpNodes &un, const int *pair_of = nullptr, const char *rebuild = nullptr,
                                      bool inline_pairs = false)
{
    un.rec.clear(); un.nvisits = 0;
    std::vector<int> edge_into(N, -1);
    for (int a = 0; a < N; a++) for (int idx = ip[a]; idx < ip[a + 1]; idx++) edge_into[ix[idx]] = idx;
    auto wanted = [&](int idx) { return !edge_mask || edge_mask[idx]; };
    /* below[a]: some edge strictly below a is wanted; sub[a]: below[a] or the edge into a */
    std::vector<char> below(N, 0), sub(N, 0);
    for (int u = N - 1; u >= 0; u--) {
        const int a = preorder[u];
        for (int idx = ip[a]; idx < ip[a + 1]; idx++) if (sub[ix[idx]]) below[a] = 1;
        sub[a] = below[a] || (edge_into[a] >= 0 && wanted(edge_into[a]));
    }
    std::vector<int> stack;
    const int root = preorder[0];
    if (ip[root + 1] > ip[root] && sub[root]) stack.push_back(root);
    int from_regs = -1;                   /* node whose G the previous visit left in registers */
    while (!stack.empty() || from_regs >= 0) {
        int a;
        if (from_regs >= 0) a = from_regs; else { a = stack.back(); stack.pop_back(); }
        const int start = ip[a], deg = ip[a + 1] - start, ea = edge_into[a];
        const int hfl = (ea >= 0 && wanted(ea) ? PLK_UN_OWN_D : 0) | (from_regs >= 0 || ea < 0 ? PLK_UN_FROM_REGS : 0);
        from_regs = -1;
        const int hdr[8] = {a, deg, node_int[a], node_scale[a], node_has_data[a] ? 1 : 0, start, ea, hfl};
        un.rec.insert(un.rec.end(), hdr, hdr + 8);
        /* inline_pairs: a pair child (two leaves, no data) is finished inside this visit: its G is used while it is in
         * registers, never stored, and the node gets no visit */
        auto inl = [&](int j) { return inline_pairs && deg <= 2 && pair_of && edge_tip[start + j] < 0 && pair_of[ix[start + j]] >= 0; };      /* (the kernel's path for more than two children has no inline form) */
        /* the child that continues in registers: the last internal child with work; its record goes last */
        int cont = -1;
        if (below[a]) for (int j = 0; j < deg; j++) if (edge_tip[start + j] < 0 && sub[ix[start + j]] && !inl(j)) cont = j;
        for (int pass = 0; pass < 2; pass++)
            for (int j = 0; j < deg; j++) {
                if ((j == cont) != (pass == 1)) continue;
                const int idx = start + j, b = ix[idx];
                const bool leaf = edge_tip[idx] >= 0;
                int fl = 0;
                if (below[a]) {
                    if (leaf) fl = wanted(idx) ? PLK_UN_LEAF_D : 0;
                    else if (inl(j)) fl = (wanted(idx) ? PLK_UN_INL_OWN : 0) | (wanted(ip[b]) ? PLK_UN_INL_L0 : 0) | (wanted(ip[b] + 1) ? PLK_UN_INL_L1 : 0);
                    else if (sub[b]) fl = j == cont ? PLK_UN_CONTINUE : PLK_UN_STORE_G;
                }
                const bool pr = !leaf && pair_of && pair_of[b] >= 0;
                const bool rbd = !leaf && !pr && rebuild && rebuild[b];
                const int cr[4] = {b, pr ? -2 - pair_of[b] : edge_tip[idx], fl | (pr ? PLK_UN_PAIR : 0) | (rbd ? PLK_UN_REBUILD : 0) | (j << PLK_UN_POS_SHIFT), leaf ? -1 : node_int[b]};
                un.rec.insert(un.rec.end(), cr, cr + 4);
            }
        un.nvisits++;
        /* stored children are visited later (depth first: the most recently stored first), the continued one next */
        for (int j = 0; j < deg; j++) if (j != cont && edge_tip[start + j] < 0 && below[a] && sub[ix[start + j]] && !inl(j)) stack.push_back(ix[start + j]);
        if (cont >= 0) from_regs = ix[start + cont];
    }
}

/* replays the kernel's walk: indices in range, every child position once, every G read after it was written (or
 * handed over in registers by the visit just before) */
static inline std::string plk_up_nodes_check(int N, int E, const int *ip, const int *ix, const PlkUpNodes &un, int nint_nodes, int ntips,
                                             int nscale_slots, int npairs = 0, const int *edge_tip = nullptr,
                                             const int *rebuild_table = nullptr, const int *pair_of = nullptr)
{
    std::vector<char> g_written(std::max(nint_nodes, 1), 0), visited(N, 0);
    size_t vp = 0;
    int handed = -1;                     /* node whose G is in registers */
    for (int v = 0; v < un.nvisits; v++) {
        if (vp + 8 > un.rec.size()) return "up nodes: record overrun";
        const int *h = &un.rec[vp];
        const int a = h[0], deg = h[1], ai = h[2], slot = h[3], e0 = h[5], ea = h[6], hfl = h[7];
        if (a < 0 || a >= N || deg < 1 || ai < 0 || ai >= nint_nodes || slot < -1 || slot >= nscale_slots) return plk_fmt("up nodes: bad header %ld", v);
        if (e0 != ip[a] || deg != ip[a + 1] - ip[a] || e0 + deg > E || ea < -1 || ea >= E || (ea >= 0 && ix[ea] != a)) return plk_fmt("up nodes: bad edge range in visit %ld", v);
        if (vp + 8 + 4 * (size_t)deg > un.rec.size()) return "up nodes: record overrun";
        if (visited[a]) return plk_fmt("up nodes: node %ld visited twice", a);
        visited[a] = 1;
        if (ea < 0) { if (v != 0 || !(hfl & PLK_UN_FROM_REGS) || handed >= 0) return "up nodes: the root's visit must be the first and take its vector from registers"; }
        else if (hfl & PLK_UN_FROM_REGS) { if (handed != a) return plk_fmt("up nodes: visit %ld expects a vector in registers that the previous visit did not leave", v); }
        else if (handed >= 0) return plk_fmt("up nodes: the vector left in registers before visit %ld is dropped", v);
        else if (!g_written[ai]) return plk_fmt("up nodes: vector of node %ld read before it is written", a);
        if (ea < 0 && (hfl & PLK_UN_OWN_D)) return "up nodes: the root has no edge";
        handed = -1;
        const int *ch = h + 8;
        std::vector<char> seen(deg, 0);
        for (int j = 0; j < deg; j++) {
            const int b = ch[4 * j], fl = ch[4 * j + 2] & ((1 << PLK_UN_POS_SHIFT) - 1), pos = ch[4 * j + 2] >> PLK_UN_POS_SHIFT, bi = ch[4 * j + 3];
            int t = ch[4 * j + 1];
            if (pos < 0 || pos >= deg || seen[pos]) return plk_fmt("up nodes: child positions of visit %ld", v);
            seen[pos] = 1;
            if (fl & PLK_UN_INL) {
                /* finished inside this visit: a pair child with nothing stored or handed on, never visited itself */
                if (!(fl & PLK_UN_PAIR) || (fl & (PLK_UN_STORE_G | PLK_UN_CONTINUE | PLK_UN_LEAF_D)) || b < 0 || b >= N) return plk_fmt("up nodes: flags of an inline pair child in visit %ld", v);
                if (visited[b]) return plk_fmt("up nodes: inline pair child %ld is also visited", b);
                visited[b] = 1;
            }
            if (fl & PLK_UN_PAIR) {
                /* the message of this child is a pair-table row: two leaf children, table index in range */
                const int pi = -2 - t;
                if (pi < 0 || pi >= npairs || b < 0 || b >= N || ip[b + 1] - ip[b] != 2) return plk_fmt("up nodes: bad pair child in visit %ld", v);
                if (edge_tip && (edge_tip[ip[b]] < 0 || edge_tip[ip[b] + 1] < 0)) return plk_fmt("up nodes: pair child with an internal child in visit %ld", v);
                t = -1;
            }
            if (fl & PLK_UN_REBUILD) {
                /* L of this child is rebuilt from its table entry: two children, each the leaf / pair node the entry names,
                 * tip slots and pair tables in range and the ones of those children */
                if ((fl & PLK_UN_PAIR) || t != -1 || !rebuild_table || !edge_tip || !pair_of || bi < 0 || bi >= nint_nodes || b < 0 || b >= N || ip[b + 1] - ip[b] != 2)
                    return plk_fmt("up nodes: bad rebuild child in visit %ld", v);
                const int *rt = rebuild_table + 4 * (size_t)bi;
                for (int q = 0; q < 2; q++) {
                    const int cn = ix[ip[b] + q], ct = edge_tip[ip[b] + q];
                    if (rt[2 * q + 1] != cn) return plk_fmt("up nodes: rebuild entry of visit %ld names the wrong child", v);
                    if (ct >= 0 ? rt[2 * q] != ct : (pair_of[cn] < 0 || pair_of[cn] >= npairs || rt[2 * q] != -2 - pair_of[cn]))
                        return plk_fmt("up nodes: rebuild entry of visit %ld names the wrong table", v);
                }
            }
            if (b != ix[e0 + pos] || t < -1 || t >= ntips) return plk_fmt("up nodes: bad child in visit %ld", v);
            if (t < 0 && (bi < 0 || bi >= nint_nodes)) return plk_fmt("up nodes: bad storage index in visit %ld", v);
            if (t >= 0 && (fl & (PLK_UN_STORE_G | PLK_UN_CONTINUE))) return plk_fmt("up nodes: a leaf gets a stored vector in visit %ld", v);
            if (t < 0 && (fl & PLK_UN_LEAF_D)) return plk_fmt("up nodes: leaf flag on an internal child in visit %ld", v);
            if ((fl & PLK_UN_CONTINUE) && (j != deg - 1 || (fl & PLK_UN_STORE_G))) return plk_fmt("up nodes: the continued child must be the last record of visit %ld", v);
            if (fl & PLK_UN_CONTINUE) handed = b;
            if (fl & PLK_UN_STORE_G) g_written[bi] = 1;
        }
        vp += 8 + 4 * (size_t)deg;
    }
    if (handed >= 0) return "up nodes: a vector is left in registers at the end";
    if (vp != un.rec.size()) return "up nodes: trailing records";
    return "";
}

static inline std::string plk_chain_check(int N, const PlkProgram &pg, const PlkChain &ch, int mode, int D, int nint_nodes,
                                          int nint_edges, int nscale_slots, int site_block, size_t lds_code_bytes)
{
    const int nops = (int)pg.ops.size(), ntips = (int)pg.tip_edge.size(), nobs = (int)pg.obs_nodes.size();
    if ((int)ch.ops.size() != nops + 1 || (ch.ops[nops].x & 0xff) != OP_END) return "down program: missing END";
    if (pg.slots_needed > D) return "down program: the tree needs a deeper stack";
    if ((size_t)nobs * site_block > lds_code_bytes) return "down program: staged code rows exceed the launch's LDS";
    std::vector<int> slot, rowv, opc;
    plk_obs_sequence(N, pg, slot, rowv, opc);
    long cur_slot = ch.first_slot, cur_row = ch.first_row;
    if (mode != 2 && !slot.empty() && (cur_slot != slot[0] || cur_row != rowv[0])) return "down program: first observation";
    if (mode != 2 && slot.empty() && ch.first_slot >= 0) return "down program: first observation without observations";
    size_t oi = 0;
    for (int pc = 0; pc < nops; pc++) {
        const plk_op4 o = ch.ops[pc];
        const int code = o.x & 0xff;
        if (code != (pg.ops[pc].x & 0xff)) return "down program: opcode mismatch";
        if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
            if (mode == 3) {
                /* two-ahead chain: y = row of the observation after next (cyclic) */
                if (o.y < 0 || o.y >= nobs || o.y != rowv[(oi + 2) % rowv.size()]) return plk_fmt("down program: op %ld prefetches the wrong row", pc);
            } else
            if (o.y < 0 || o.y >= nobs || o.y != rowv[oi]) return plk_fmt("down program: op %ld names the wrong staged row", pc);
            if (code != OP_NODE_MUL && ((o.x >> 8) < 0 || (o.x >> 8) >= ntips)) return "down program: tip slot out of range";
            if (mode != 2) {
                const int wrap = (o.z >> 30) & 1, ns = o.z & 0x3fffffff;
                if (ns < 0 || ns > ntips || o.w < 0 || o.w >= nobs) return plk_fmt("down program: op %ld prefetches out of range", pc);
                if (cur_slot != slot[oi] || cur_row != rowv[oi]) return plk_fmt("down program: chain delivers the wrong observation at op %ld", pc);
                const size_t nx = oi + 1 < slot.size() ? oi + 1 : 0;
                if (wrap != (oi + 1 == slot.size()) || ns != slot[nx] || o.w != rowv[nx]) return plk_fmt("down program: chain link after op %ld", pc);
                cur_slot = ns; cur_row = o.w;
            }
            oi++;
        } else if (code == OP_MATVEC && mode == 0) {
            /* z = the next MATVEC's op index, cyclically */
            int nx = (int)pc;
            do { nx = nx + 1 < nops ? nx + 1 : 0; } while ((pg.ops[nx].x & 0xff) != OP_MATVEC);
            if (o.z != nx) return plk_fmt("down program: op %ld names the wrong next product", pc);
        } else if (code == OP_MATVEC && mode >= 1) {
            if (o.y != pg.op_edge[pc] || o.z < (mode >= 1 && nint_nodes > 0 ? -1 : 0) || o.z >= (nint_nodes > 0 ? nint_nodes : nops)) return plk_fmt("down program: op %ld stores to a bad node index", pc);
            if (mode == 2 && (o.w < 0 || o.w >= nint_edges)) return plk_fmt("down program: op %ld stores to a bad edge index", pc);
        } else if (code == OP_PUSH || code == OP_POPMUL) {
            if (o.y < 0 || o.y >= D) return "down program: stack slot out of range";
        } else if (code == OP_SCALE && mode >= 1 && nscale_slots > 0) {
            if (o.y < -1 || o.y >= nscale_slots) return "down program: rescaling slot out of range";
        }
    }
    if (oi != slot.size()) return "down program: observation count";
    if (mode == 3 && !rowv.empty() && ch.second_row != rowv[1 % rowv.size()]) return "down program: second row";
    return "";
}

#endif
