/*
 * tests/progcheck_main.cpp -- CPU driver for the traversal-program builder and the host replays of the device
 * interpreters (phyly_amd/csrc/plk_program.h), built with AddressSanitizer + UBSan by tests/test_program_check.py.
 *
 * usage: progcheck                      seeded random trees of every shape the engine accepts
 *        progcheck <file>               trees from a file, one per line: "N nchar  a0 b0 a1 b1 ... | h0 h1 ... hN-1"
 *                                       (edges parent child in user order; h = 1 for nodes that carry data)
 * For every tree: program build + invariants, fused formats for every kernel variant the engine could launch
 * (assembly interpreter with 4-bit and 8-bit codes, C++ interpreter with one and two sites per lane), the chained
 * programs of the three down passes; and negative controls (corrupted words must be refused).
 * Prints "ok <trees> <ops>" and exits 0, or the first failure and exits 1.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "plk_program.h"

struct Tree { int N; std::vector<int> ip, ix, pre; };

/* CSR + BFS order as host_model.c builds them (src/csr_graph.c:218-229 of the reference: children in user order) */
static bool make_tree(int N, const std::vector<int> &ea, const std::vector<int> &eb, Tree &t)
{
    t.N = N; t.ip.assign(N + 1, 0); t.ix.assign(N > 1 ? N - 1 : 1, 0); t.pre.clear();
    std::vector<int> indeg(N, 0);
    for (int e = 0; e < N - 1; e++) { t.ip[ea[e] + 1]++; indeg[eb[e]]++; }
    for (int a = 0; a < N; a++) t.ip[a + 1] += t.ip[a];
    std::vector<int> fill(t.ip.begin(), t.ip.end() - 1);
    for (int e = 0; e < N - 1; e++) t.ix[fill[ea[e]]++] = eb[e];
    int root = -1;
    for (int a = 0; a < N; a++) if (!indeg[a]) { if (root >= 0) return false; root = a; }
    if (root < 0) return false;
    t.pre.push_back(root);
    for (size_t h = 0; h < t.pre.size(); h++)
        for (int idx = t.ip[t.pre[h]]; idx < t.ip[t.pre[h] + 1]; idx++) t.pre.push_back(t.ix[idx]);
    return (int)t.pre.size() == N;
}

static long g_ops = 0;

static std::string check_tree(const Tree &t, const std::vector<char> &has, int nchar)
{
    const int N = t.N;
    PlkProgram pg;
    plk_program_build(N, t.ip.data(), t.ix.data(), t.pre.data(), has.data(), pg);
    g_ops += (long)pg.ops.size();
    std::string bad = plk_program_check(N, t.ip.data(), t.ix.data(), t.pre.data(), has.data(), pg);
    if (!bad.empty()) return bad;
    /* storage indices as run_updown4 / run_updown_mfma compute them */
    const int E = N - 1, ntips = (int)pg.tip_edge.size();
    std::vector<int> edge_tip(E, -1), edge_int(E, -1), node_int(N, -1), node_scale(N, -1);
    for (int k = 0; k < ntips; k++) edge_tip[pg.tip_edge[k]] = k;
    int nie = 0, nin = 0, nsc = 0;
    for (int e = 0; e < E; e++) if (edge_tip[e] < 0) edge_int[e] = nie++;
    for (int a = 0; a < N; a++) if (t.ip[a + 1] > t.ip[a]) node_int[a] = nin++;
    for (int a = 0; a < N; a++) if (node_int[a] >= 0 && pg.scale_node[a]) node_scale[a] = nsc++;
    for (int mode = 0; mode <= 4; mode++) {
        PlkChain ch;
        if (mode == 4) {          /* mode 3 without node storage (the vector ll kernel) */
            plk_chain_build(N, pg, 3, t.ix.data(), nullptr, nullptr, nullptr, ch);
            bad = plk_chain_check(N, pg, ch, 3, 1 << 30, 0, 0, 0, 0, 0);
            if (!bad.empty()) return "mode 3 (ll): " + bad;
            continue;
        }
        plk_chain_build(N, pg, mode, t.ix.data(), node_int.data(), edge_int.data(), node_scale.data(), ch);
        const int block = mode == 1 ? 256 : 64;
        bad = plk_chain_check(N, pg, ch, mode, 1 << 30, nin, nie, nsc, block, pg.obs_nodes.size() * (size_t)block);
        if (!bad.empty()) return "mode " + std::to_string(mode) + ": " + bad;
        if (mode == 1 && !pg.ops.empty()) {      /* negative control: a broken chain link must be noticed */
            for (size_t pc = 0; pc < pg.ops.size(); pc++) {
                const int code = ch.ops[pc].x & 0xff;
                if (code == OP_TIP_SET || code == OP_TIP_MUL || code == OP_NODE_MUL) {
                    PlkChain c2 = ch;
                    c2.ops[pc].w += 1;
                    if (plk_chain_check(N, pg, c2, mode, 1 << 30, nin, nie, nsc, block, pg.obs_nodes.size() * (size_t)block).empty())
                        return "negative control: corrupted chain accepted";
                    break;
                }
            }
        }
    }
    /* visit records + matrix stream of the vector up pass, for every kind of query and with random masks */
    {
        std::vector<int> emask(E > 0 ? E : 1), nmask(N);
        unsigned lcg = 12345u + (unsigned)N * 7919u;
        for (int e = 0; e < E; e++) { lcg = lcg * 1664525u + 1013904223u; emask[e] = (lcg >> 24) % 3 != 0; }
        for (int a = 0; a < N; a++) { lcg = lcg * 1664525u + 1013904223u; nmask[a] = (lcg >> 24) % 3 != 0; }
        const int modes[3][2] = {{1, 0}, {0, 1}, {1, 1}};
        /* observed[a]: every site of leaf a is a single state (marginal queries then inline pre-terminal nodes too): none, a
         * pseudo-random half, all */
        std::vector<char> observed(N, 0);
        for (int md = 0; md < 3; md++)
            for (int masked = 0; masked < 2; masked++)
            for (int ob = 0; ob < (modes[md][1] ? 3 : 1); ob++) {
                for (int a = 0; a < N; a++) { lcg = lcg * 1664525u + 1013904223u; observed[a] = ob == 2 || (ob == 1 && (lcg >> 24) % 2); }
                PlkUpVisits uv;
                plk_up_visits_build(N, t.ip.data(), t.ix.data(), t.pre.data(), has.data(), edge_tip.data(), node_int.data(),
                                    node_scale.data(), modes[md][0] != 0, modes[md][1] != 0, masked ? emask.data() : nullptr,
                                    masked ? nmask.data() : nullptr, uv, ob ? observed.data() : nullptr);
                bad = plk_up_visits_check(N, E, uv, nin, ntips, nsc, modes[md][0] != 0);
                if (!bad.empty()) return "up visits (deriv " + std::to_string(modes[md][0]) + ", marg " + std::to_string(modes[md][1]) + "): " + bad;
                if (modes[md][1]) {
                    /* every wanted marginal is produced exactly once: by a visit's child record, by the root's header, or by
                     * an inline record's leaf entry; an inlined node's leaves are all observed */
                    std::vector<int> mdone(N, 0);
                    size_t vp = 0;
                    for (int v = 0; v < uv.nvisits; v++) {
                        const int *h = &uv.rec[vp];
                        if (h[6]) mdone[h[0]]++;
                        for (int j = 0; j < h[1]; j++) {
                            const int *c = h + 8 + 4 * j;
                            if (c[2] & PLK_UP_WANT_M) mdone[c[0]]++;
                            if (c[2] & PLK_UP_INLINE) {
                                const int *q = &uv.rec[-2 - c[1]];
                                for (int l = 0; l < q[3]; l++) {
                                    if (q[7 + 3 * l] & 2) mdone[q[5 + 3 * l]]++;
                                    if (!ob || !observed[q[5 + 3 * l]]) return "up visits: a node with an unobserved leaf is inlined in a marginal pass";
                                }
                            }
                        }
                        vp += 8 + 4 * (size_t)h[1];
                    }
                    for (int a = 0; a < N; a++)
                        if (mdone[a] != ((!masked || nmask[a]) ? 1 : 0)) return "up visits: marginal of node " + std::to_string(a) + " produced " + std::to_string(mdone[a]) + " times";
                }
                if (md == 0 && !masked && !uv.kind.empty()) {      /* negative control: a dropped matrix must be noticed */
                    PlkUpVisits u2 = uv;
                    u2.kind.pop_back(); u2.edge.pop_back();
                    if (plk_up_visits_check(N, E, u2, nin, ntips, nsc, true).empty()) return "negative control: short matrix stream accepted";
                }
            }
    }
    /* node visits of the derivative up pass (k_up_nodes_mfma): every wanted edge is finished exactly once, and every
     * edge that is finished is wanted */
    if (E > 0) {
        std::vector<int> emask(E);
        unsigned lcg = 999u + (unsigned)N * 31u;
        for (int e = 0; e < E; e++) { lcg = lcg * 1664525u + 1013904223u; emask[e] = (lcg >> 24) % 4 == 0; }
        for (int masked = 0; masked < 2; masked++) {
            PlkUpNodes un;
            plk_up_nodes_build(N, t.ip.data(), t.ix.data(), t.pre.data(), has.data(), edge_tip.data(), node_int.data(),
                               node_scale.data(), masked ? emask.data() : nullptr, un);
            bad = plk_up_nodes_check(N, E, t.ip.data(), t.ix.data(), un, nin, ntips, nsc);
            if (!bad.empty()) return std::string("up nodes (") + (masked ? "masked" : "all edges") + "): " + bad;
            std::vector<int> done(E, 0);
            size_t vp = 0;
            for (int v = 0; v < un.nvisits; v++) {
                const int *h = &un.rec[vp];
                if ((h[7] & PLK_UN_OWN_D) && h[6] >= 0) done[h[6]]++;
                for (int j = 0; j < h[1]; j++) if (h[8 + 4 * j + 2] & PLK_UN_LEAF_D) done[h[5] + (h[8 + 4 * j + 2] >> PLK_UN_POS_SHIFT)]++;
                vp += 8 + 4 * (size_t)h[1];
            }
            for (int e = 0; e < E; e++)
                if (done[e] != ((!masked || emask[e]) ? 1 : 0)) return "up nodes: edge " + std::to_string(e) + " finished " + std::to_string(done[e]) + " times";
            if (!masked) {                 /* negative control: drop the first stored vector, its reader must be caught */
                PlkUpNodes u2 = un;
                for (size_t q = 0, w = 0; w < (size_t)u2.nvisits; w++) {
                    const int dg = u2.rec[q + 1];
                    bool hit = false;
                    for (int j = 0; j < dg; j++) if (u2.rec[q + 8 + 4 * j + 2] & PLK_UN_STORE_G) { u2.rec[q + 8 + 4 * j + 2] &= ~PLK_UN_STORE_G; hit = true; break; }
                    if (hit) { if (plk_up_nodes_check(N, E, t.ip.data(), t.ix.data(), u2, nin, ntips, nsc).empty()) return "negative control: unwritten vector accepted (up nodes)"; break; }
                    q += 8 + 4 * (size_t)dg;
                }
            }
        }
    }
    /* the k = 4 node visits with pair tables and table-rebuilt children (run_updown4): pair nodes as the engine picks them,
     * rebuild table, records, and a negative control (an entry that names the wrong table must be refused) */
    if (E > 0) {
        std::vector<int> pair_of(N, -1), edge_into(N, -1);
        for (int a = 0; a < N; a++) for (int idx = t.ip[a]; idx < t.ip[a + 1]; idx++) edge_into[t.ix[idx]] = idx;
        int npairs = 0;
        for (int b = 0; b < N; b++) {
            const int e0 = t.ip[b];
            if (t.ip[b + 1] - e0 != 2 || edge_into[b] < 0 || has[b]) continue;
            if (edge_tip[e0] < 0 || edge_tip[e0 + 1] < 0) continue;
            pair_of[b] = npairs++;
        }
        std::vector<char> rb;
        std::vector<int> rtab;
        plk_up_rebuild_table(N, t.ip.data(), t.ix.data(), has.data(), edge_tip.data(), node_int.data(), node_scale.data(), pair_of.data(), nin, rb, rtab);
        PlkUpNodes un;
        plk_up_nodes_build(N, t.ip.data(), t.ix.data(), t.pre.data(), has.data(), edge_tip.data(), node_int.data(), node_scale.data(), nullptr, un,
                           pair_of.data(), rb.data());
        bad = plk_up_nodes_check(N, E, t.ip.data(), t.ix.data(), un, nin, ntips, nsc, npairs, edge_tip.data(), rtab.data(), pair_of.data());
        if (!bad.empty()) return "up nodes (pairs, rebuilt children): " + bad;
        /* pair children finished inside their parent's visit, without and with an edge mask: every wanted edge still exactly once */
        {
            std::vector<int> emask(E);
            unsigned lcg2 = 4242u + (unsigned)N * 17u;
            for (int e = 0; e < E; e++) { lcg2 = lcg2 * 1664525u + 1013904223u; emask[e] = (lcg2 >> 24) % 3 != 0; }
            for (int masked = 0; masked < 2; masked++) {
                PlkUpNodes ui;
                plk_up_nodes_build(N, t.ip.data(), t.ix.data(), t.pre.data(), has.data(), edge_tip.data(), node_int.data(), node_scale.data(),
                                   masked ? emask.data() : nullptr, ui, pair_of.data(), rb.data(), true);
                bad = plk_up_nodes_check(N, E, t.ip.data(), t.ix.data(), ui, nin, ntips, nsc, npairs, edge_tip.data(), rtab.data(), pair_of.data());
                if (!bad.empty()) return "up nodes (inline pair children): " + bad;
                std::vector<int> done(E, 0);
                size_t vq = 0;
                for (int v = 0; v < ui.nvisits; v++) {
                    const int *h = &ui.rec[vq];
                    if ((h[7] & PLK_UN_OWN_D) && h[6] >= 0) done[h[6]]++;
                    for (int j = 0; j < h[1]; j++) {
                        const int fl = h[8 + 4 * j + 2], b = h[8 + 4 * j];
                        if (fl & PLK_UN_LEAF_D) done[h[5] + (fl >> PLK_UN_POS_SHIFT)]++;
                        if (fl & PLK_UN_INL_OWN) done[h[5] + (fl >> PLK_UN_POS_SHIFT)]++;
                        if (fl & PLK_UN_INL_L0) done[t.ip[b]]++;
                        if (fl & PLK_UN_INL_L1) done[t.ip[b] + 1]++;
                        if ((fl & PLK_UN_INL) && h[1] > 2) return "up nodes: inline pair child under a node with more than two children";
                    }
                    vq += 8 + 4 * (size_t)h[1];
                }
                for (int e = 0; e < E; e++)
                    if (done[e] != ((!masked || emask[e]) ? 1 : 0)) return "up nodes (inline pairs): edge " + std::to_string(e) + " finished " + std::to_string(done[e]) + " times";
            }
        }
        int nflag = 0, nrb = 0;
        size_t vp = 0;
        for (int v = 0; v < un.nvisits; v++) {
            const int *h = &un.rec[vp];
            for (int j = 0; j < h[1]; j++) if (h[8 + 4 * j + 2] & PLK_UN_REBUILD) nflag++;
            vp += 8 + 4 * (size_t)h[1];
        }
        for (int b = 0; b < N; b++) if (rb[b]) {
            nrb++;
            if (pair_of[b] >= 0 || has[b] || t.ip[b + 1] - t.ip[b] != 2) return "rebuild table: node " + std::to_string(b) + " does not qualify";
        }
        if (nflag != nrb) return "up nodes: " + std::to_string(nrb) + " rebuilt nodes, " + std::to_string(nflag) + " flagged child records";
        if (nrb > 0) {
            std::vector<int> r2 = rtab;
            for (int b = 0; b < N; b++) if (rb[b]) { r2[4 * (size_t)node_int[b]] += r2[4 * (size_t)node_int[b]] >= 0 ? 1 : -1; break; }
            if (plk_up_nodes_check(N, E, t.ip.data(), t.ix.data(), un, nin, ntips, nsc, npairs, edge_tip.data(), r2.data(), pair_of.data()).empty())
                return "negative control: rebuild entry with the wrong table accepted";
        }
    }
    if (pg.slots_needed > PLK_FUSED_SLOTS) return "";
    PlkFused fu;
    plk_fused_build(N, pg, fu);
    for (int NS = 1; NS <= 2; NS++) {
        const size_t lds = plk_fused_lds_bytes(pg, nchar, PLK_TILE * NS);
        if (lds > PLK_LDS_LIMIT || (NS == 2 && pg.slots_needed > 8)) continue;
        const int D = pg.slots_needed <= 4 ? 4 : (pg.slots_needed <= 8 ? 8 : 16);
        bad = plk_fused_check_cpp(N, pg, fu, nchar, D, NS, lds);
        if (!bad.empty()) return "NS " + std::to_string(NS) + ": " + bad;
    }
    if (fu.asm_ok) {
        for (int pack4 = 0; pack4 <= 1; pack4++) {
            if (pack4 && nchar > 16) continue;
            const size_t lds = plk_fused_lds_bytes(pg, nchar, pack4 ? PLK_TILE / 2 : PLK_TILE);
            if (lds > PLK_LDS_LIMIT) continue;
            const int D = pg.slots_needed <= 4 ? 4 : 8;
            bad = plk_fused_check_asm(N, pg, fu, nchar, D, pack4, lds);
            if (!bad.empty()) return "pack4 " + std::to_string(pack4) + ": " + bad;
            /* negative controls: one byte less of LDS, a field pushed out of range, a lost END */
            if (plk_fused_check_asm(N, pg, fu, nchar, D, pack4, lds - 1).empty()) return "negative control: short LDS accepted";
            PlkFused f2 = fu;
            bool touched = false;
            for (size_t w = 0; w < f2.words.size() && !touched; w++)
                if ((f2.words[w] & 31) <= 1) { f2.words[w] = (f2.words[w] & 0xffff) | ((unsigned)pg.obs_nodes.size() << 16); touched = true; }
            if (touched && plk_fused_check_asm(N, pg, f2, nchar, D, pack4, lds).empty()) return "negative control: row field out of range accepted";
            f2 = fu;
            for (size_t w = 0; w < f2.words.size(); w++)
                if ((f2.words[w] & 31) == OP_END) { f2.words[w] = OP_SCALE; break; }      /* the first END overwritten */
            if (plk_fused_check_asm(N, pg, f2, nchar, D, pack4, lds).empty()) return "negative control: missing END accepted";
        }
    }
    /* the vector ll kernel's program on pair tables (any stack depth, any nchar): every cherry, a budget of two */
    if (!pg.obs_nodes.empty()) {
        const int vbud[2] = {1 << 30, 2};
        for (int bi = 0; bi < 2; bi++) {
            PlkFusedPT fp;
            plk_fused_pt_build(N, t.ip.data(), t.ix.data(), pg, nchar, vbud[bi], fp, false);
            PlkVecPT vp;
            plk_vec_pt_build(fp, nchar, vp);
            bad = plk_vec_pt_check(N, t.ip.data(), t.ix.data(), pg, fp, vp, nchar, E);
            if (!bad.empty()) return "vec pt budget " + std::to_string(bi) + ": " + bad;
            if (bi == 0 && fp.npairs > 0) {           /* negative controls */
                PlkVecPT v2 = vp;
                for (size_t q = 0; q < v2.ops.size(); q++) if ((v2.ops[q].x & 0xff) == OP_TIP_SET) { v2.ops[q].y += nchar; break; }
                if (plk_vec_pt_check(N, t.ip.data(), t.ix.data(), pg, fp, v2, nchar, E).empty()) return "negative control (vec pt): shifted table accepted";
                v2 = vp;
                bool changed = false;
                for (size_t q = 0; q < v2.op_edge.size() && !changed; q++) if (v2.op_edge[q] >= 0) { v2.op_edge[q] = (v2.op_edge[q] + 1) % (E > 0 ? E : 1); changed = true; }
                if (changed && E > 1 && plk_vec_pt_check(N, t.ip.data(), t.ix.data(), pg, fp, v2, nchar, E).empty()) return "negative control (vec pt): wrong matrix accepted";
            }
        }
    }
    /* pair-table interpreter (k_ll_fused4_asm_pt): with every cherry as a table, with a budget of one, with none */
    if (pg.slots_needed <= 4 && nchar <= 16 && !pg.obs_nodes.empty()) {
        const int budgets[3] = {1 << 30, 1, 0};
        for (int bi = 0; bi < 3; bi++) {
            PlkFusedPT fp;
            plk_fused_pt_build(N, t.ip.data(), t.ix.data(), pg, nchar, budgets[bi], fp);
            if (fp.npairs > budgets[bi]) return "pt: more pair tables than the budget";
            const int tile = (N + bi) % 2 ? 512 : 1024;
            const size_t lds = plk_fused_pt_lds_bytes(fp, nchar, tile);
            if (lds > plk_pt_lds_limit(tile) || fp.units >= 2048) continue;
            bad = plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, fp, nchar, tile, lds);
            if (!bad.empty()) return "pt budget " + std::to_string(bi) + ": " + bad;
            /* every edge is either a matrix of the stream, the leaf edge of a one-unit table, or one of the three edges
             * of a pair table -- exactly once */
            std::vector<int> seen(E > 0 ? E : 1, 0);
            for (int e : fp.mat_edge) seen[e]++;
            for (size_t q = 0; q < fp.tab_edge.size(); q++) {
                if (fp.tab_edge[q] >= 0) seen[fp.tab_edge[q]]++;
                if (fp.tab_eb[q] >= 0) { seen[fp.tab_eb[q]]++; seen[fp.tab_ec[q]]++; }
            }
            for (int e = 0; e < E; e++) if (seen[e] != 1) return "pt: edge " + std::to_string(e) + " covered " + std::to_string(seen[e]) + " times";
            {
                /* the two-sites-per-lane interpreter's words: TIP_SET + POPMUL / PUSH fused, re-encoded as 64-bit ops */
                PlkFusedPT fq;
                plk_fused_pt_build(N, t.ip.data(), t.ix.data(), pg, nchar, budgets[bi], fq, true);
                const int tile4 = (N + bi) % 2 ? 1024 : 1536;
                const size_t lds4 = plk_fused_pt_lds_bytes(fq, nchar, tile4);
                if (lds4 <= plk_pt_lds_limit(tile4) && fq.units < 2048) {
                    bad = plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, fq, nchar, tile4, lds4, true);
                    if (!bad.empty()) return "v4 words, budget " + std::to_string(bi) + ": " + bad;
                    if (fq.words != fp.words && plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, fq, nchar, tile4, lds4, false).empty())
                        return "negative control (v4): fused SET words accepted by the one-site interpreter's check";
                    PlkFusedV4 v4;
                    plk_fused_v4_words(fq, nchar, tile4, 256, v4);
                    bad = plk_fused_check_v4(fq, v4, nchar, tile4, 256, lds4);
                    if (!bad.empty()) return "v4 encoding: " + bad;
                    PlkFusedV4 v5 = v4;
                    v5.words[1] += 1u << 16;
                    if (plk_fused_check_v4(fq, v5, nchar, tile4, 256, lds4).empty() && plk_word_is_obs(fq.words[0] & 31)) return "negative control (v4): shifted table offset accepted";
                }
            }
            if (bi != 0) continue;
            /* negative controls */
            if (plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, fp, nchar, tile, lds - 1).empty()) return "negative control (pt): short LDS accepted";
            PlkFusedPT f2 = fp;
            for (size_t w = 0; w < f2.words.size(); w++)
                if ((f2.words[w] & 31) == OP_END) { f2.words[w] = OP_SCALE; break; }
            if (plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, f2, nchar, tile, lds).empty()) return "negative control (pt): missing END accepted";
            if (fp.npairs > 0) {
                f2 = fp;
                for (size_t q = 0; q < f2.tab_eb.size(); q++) if (f2.tab_eb[q] >= 0) { std::swap(f2.tab_eb[q], f2.tab_edge[q]); break; }
                if (plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, f2, nchar, tile, lds).empty()) return "negative control (pt): pair table with swapped edges accepted";
                f2 = fp;
                for (size_t r = 0; r < f2.row_node2.size(); r++) if (f2.row_node2[r] >= 0) { f2.row_node2[r] = -1; break; }
                if (plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, f2, nchar, tile, lds).empty()) return "negative control (pt): pair row without its second node accepted";
            }
            f2 = fp;
            bool touched = false;
            for (size_t w = 0; w < f2.words.size() && !touched; w++)
                if ((f2.words[w] & 31) <= 1) { f2.words[w] = (f2.words[w] & 0xffff) | ((unsigned)fp.row_node.size() << 16); touched = true; }
            if (touched && plk_fused_check_pt(N, t.ip.data(), t.ix.data(), pg, f2, nchar, tile, lds).empty()) return "negative control (pt): row field out of range accepted";
        }
    }
    return "";
}

int main(int argc, char **argv)
{
    long ntrees = 0;
    if (argc > 1) {
        FILE *f = fopen(argv[1], "r");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
        int N, nchar;
        while (fscanf(f, "%d %d", &N, &nchar) == 2) {
            if (N < 2) return 2;
            std::vector<int> ea(N - 1), eb(N - 1);
            for (int e = 0; e < N - 1; e++) if (fscanf(f, "%d %d", &ea[e], &eb[e]) != 2) return 2;
            char bar[4];
            if (fscanf(f, "%3s", bar) != 1 || bar[0] != '|') return 2;
            std::vector<char> has(N);
            for (int a = 0; a < N; a++) { int v; if (fscanf(f, "%d", &v) != 1) return 2; has[a] = (char)v; }
            Tree t;
            if (!make_tree(N, ea, eb, t)) { fprintf(stderr, "tree %ld is not a rooted tree\n", ntrees); return 1; }
            for (int a = 0; a < N; a++) if (t.ip[a + 1] == t.ip[a]) has[a] = 0;   /* leaves are tips, not NODE_MUL */
            const std::string bad = check_tree(t, has, nchar);
            if (!bad.empty()) { fprintf(stderr, "tree %ld (N = %d, nchar = %d): %s\n", ntrees, N, nchar, bad.c_str()); return 1; }
            ntrees++;
        }
        fclose(f);
        printf("ok %ld %ld\n", ntrees, g_ops);
        return 0;
    }
    std::mt19937_64 rng(20250355);
    auto rnd = [&](int n) { return (int)(rng() % (unsigned long long)n); };
    const int nchars[] = {1, 2, 5, 16, 17, 64, 255, 256};
    for (int iter = 0; iter < 6000; iter++) {
        const int shape = iter % 8;
        int N;
        if (shape == 7) N = 2 + rnd(3);
        else if (iter % 97 == 0) N = 500 + rnd(3000);
        else N = 2 + rnd(iter % 5 == 0 ? 300 : 40);
        std::vector<int> ea(N - 1), eb(N - 1), label(N);
        for (int i = 0; i < N; i++) label[i] = i;
        for (int i = N - 1; i > 0; i--) std::swap(label[i], label[rnd(i + 1)]);
        for (int i = 1; i < N; i++) {
            int parent;
            switch (shape) {
            case 0: parent = rnd(i); break;                                   /* random recursive tree (multifurcating) */
            case 1: parent = i - 1; break;                                    /* chain: every node has one child */
            case 2: parent = (i - 1) / 2; break;                              /* complete binary */
            case 3: parent = i % 2 ? i - 1 - (i > 1) : i - 2; if (parent < 0) parent = 0; break;   /* caterpillar */
            case 4: parent = 0; break;                                        /* star */
            case 5: parent = rnd(10) < 7 ? rnd(i) : std::max(0, i - 1 - rnd(std::min(i, 3))); break;  /* the differential generator's mix */
            case 6: parent = (i - 1) / 3; break;                              /* ternary */
            default: parent = rnd(i); break;
            }
            ea[i - 1] = label[parent]; eb[i - 1] = label[i];
        }
        for (int e = N - 2; e > 0; e--) { const int j = rnd(e + 1); std::swap(ea[e], ea[j]); std::swap(eb[e], eb[j]); }
        Tree t;
        if (!make_tree(N, ea, eb, t)) { fprintf(stderr, "generator produced a non-tree\n"); return 2; }
        std::vector<char> has(N, 0);
        const int pdata = iter % 3 == 0 ? 0 : (iter % 3 == 1 ? 30 : 100);
        for (int a = 0; a < N; a++) has[a] = t.ip[a + 1] > t.ip[a] && rnd(100) < pdata;
        const std::string bad = check_tree(t, has, nchars[rnd(8)]);
        if (!bad.empty()) { fprintf(stderr, "iteration %d (shape %d, N = %d): %s\n", iter, shape, N, bad.c_str()); return 1; }
        ntrees++;
    }
    /* deep stacks: complete binary trees up to 4096 leaves (stack depth beyond every register variant) */
    for (int d = 2; d <= 12; d++) {
        const int N = (1 << (d + 1)) - 1;
        std::vector<int> ea(N - 1), eb(N - 1);
        for (int i = 1; i < N; i++) { ea[i - 1] = (i - 1) / 2; eb[i - 1] = i; }
        Tree t;
        make_tree(N, ea, eb, t);
        std::vector<char> has(N, 0);
        const std::string bad = check_tree(t, has, 5);
        if (!bad.empty()) { fprintf(stderr, "balanced depth %d: %s\n", d, bad.c_str()); return 1; }
        ntrees++;
    }
    printf("ok %ld %ld\n", ntrees, g_ops);
    return 0;
}
