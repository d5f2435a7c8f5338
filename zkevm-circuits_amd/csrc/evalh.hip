// `plonk::evaluation::Evaluator::evaluate_h` on gfx950: quotient numerator over the extended coset.
//
// One lane per extended row.  Custom gates and each lookup's compressed-expression graph arrive as the
// serialised `GraphEvaluator`; the host lowers a graph to a register-machine program:
//   * only calculations reachable from the result are emitted, in just-in-time order, so a value is computed
//     right before its first use instead of all "parts" being alive until the final Horner;
//   * `Store` of a column / challenge is never materialised: users read the source directly;
//   * `Horner(start, parts, factor)` becomes one multiply-add per part on an accumulator that lives in
//     registers for the root calculation;
//   * values get slots by liveness; the slots with the densest use live in LDS ([slot][lane], 32 B each), the
//     rest spill to a [slot][row] buffer in HBM (coalesced 32 B per lane).
// The program is wave-uniform, so decode runs on the scalar unit and every branch is a scalar branch; column
// reads are (row + rotation * 2^(ext_k - k)) mod 2^ext_k, i.e. coalesced 32-byte loads per lane.
// Permutation and lookup argument terms are fixed-function code following the upstream formulas.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "domain.h"

using namespace zkhip;

namespace zkhip {

constexpr int EV_THREADS = 256;

enum DevOp : uint32_t { D_ADD = 0, D_SUB, D_MUL, D_SQR, D_DBL, D_NEG, D_MOV, D_MULADD };
enum DevKind : uint32_t { K_CONST = 0, K_SLOT, K_COL, K_SCALAR, K_ACC, K_ZERO, K_PREV };
constexpr uint32_t DST_ACC = 0xffffffu;

// scalar table layout: [0] beta [1] gamma [2] theta [3] y [4..] challenges
constexpr uint32_t SC_BETA = 0, SC_GAMMA = 1, SC_THETA = 2, SC_Y = 3, SC_CHALLENGE0 = 4;

struct InterpArgs {
    const uint32_t* prog;       // 4 words per instruction
    uint32_t n_instr;
    uint32_t result;            // operand word of the graph's value
    const fe* consts;
    const fe* scalars;
    const fe* const* cols;      // unified column table: fixed | advice | instance (extended cosets)
    const int32_t* rots;
    uint32_t lds_slots;
    fe* spill;                  // [slot - lds_slots][size]
    uint32_t log_size;          // extended_k
    uint32_t rot_scale;         // 2^(extended_k - k)
    fe* out;                    // values[], updated in place
    // lookup mode (mode = 1)
    uint32_t mode;
    const fe* product;
    const fe* permuted_input;
    const fe* permuted_table;
    const fe* l0;
    const fe* l_last;
    const fe* l_active;
};

ZK_DEV uint32_t rot_row(uint32_t idx, int32_t rot, uint32_t rot_scale, uint32_t mask) {
    return (uint32_t)((int32_t)idx + rot * (int32_t)rot_scale) & mask;
}

__device__ __forceinline__ fe ev_fetch(const InterpArgs& a, uint32_t word, const fe* lds, const fe& acc, const fe& prev,
                                       uint32_t idx, uint32_t mask) {
    const uint32_t kind = word >> 28, pl = word & 0x0fffffffu;
    switch (kind) {
        case K_CONST: return load_fe(a.consts + pl);
        case K_SLOT:
            if (pl < a.lds_slots) return lds[pl * EV_THREADS + threadIdx.x];
            return load_fe(a.spill + ((size_t)(pl - a.lds_slots) << a.log_size) + idx);
        case K_COL: {
            const fe* col = a.cols[pl >> 12];
            const int32_t rot = a.rots[pl & 0xfff];
            return load_fe(col + rot_row(idx, rot, a.rot_scale, mask));
        }
        case K_SCALAR: return load_fe(a.scalars + pl);
        case K_ACC: return acc;
        case K_PREV: return prev;
        default: return Fr::zero();
    }
}

__global__ void __launch_bounds__(EV_THREADS) evalh_interp_kernel(const InterpArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    fe* lds = reinterpret_cast<fe*>(smem);
    const uint32_t idx = blockIdx.x * EV_THREADS + threadIdx.x;
    const uint32_t mask = (1u << a.log_size) - 1;
    if (idx > mask) return;   // domains smaller than one workgroup
    const fe prev = load_fe(a.out + idx);
    const fe pv = (a.mode == 0) ? prev : Fr::zero();   // `previous_value` of GraphEvaluator::evaluate
    fe acc = pv;
    for (uint32_t pc = 0; pc < a.n_instr; pc++) {
        const uint32_t w0 = __builtin_amdgcn_readfirstlane(a.prog[4 * pc]);
        const uint32_t wa = __builtin_amdgcn_readfirstlane(a.prog[4 * pc + 1]);
        const uint32_t wb = __builtin_amdgcn_readfirstlane(a.prog[4 * pc + 2]);
        const uint32_t wc = __builtin_amdgcn_readfirstlane(a.prog[4 * pc + 3]);
        const uint32_t op = w0 & 0xff, dst = w0 >> 8;
        const fe x = ev_fetch(a, wa, lds, acc, pv, idx, mask);
        fe r;
        switch (op) {
            case D_ADD: r = Fr::add(x, ev_fetch(a, wb, lds, acc, pv, idx, mask)); break;
            case D_SUB: r = Fr::sub(x, ev_fetch(a, wb, lds, acc, pv, idx, mask)); break;
            case D_MUL: r = Fr::mul(x, ev_fetch(a, wb, lds, acc, pv, idx, mask)); break;
            case D_SQR: r = Fr::sqr(x); break;
            case D_DBL: r = Fr::dbl(x); break;
            case D_NEG: r = Fr::neg(x); break;
            case D_MOV: r = x; break;
            default: {   // D_MULADD: x * b + c
                const fe y = ev_fetch(a, wb, lds, acc, pv, idx, mask);
                r = Fr::add(Fr::mul(x, y), ev_fetch(a, wc, lds, acc, pv, idx, mask));
                break;
            }
        }
        if (dst == DST_ACC) acc = r;
        else if (dst < a.lds_slots) lds[dst * EV_THREADS + threadIdx.x] = r;
        else store_fe(a.spill + ((size_t)(dst - a.lds_slots) << a.log_size) + idx, r);
    }
    const fe value = ev_fetch(a, a.result, lds, acc, pv, idx, mask);
    if (a.mode == 0) {
        store_fe(a.out + idx, value);
        return;
    }
    // lookup argument terms; `value` = (theta-compressed input + beta) * (theta-compressed table + gamma)
    const fe y = load_fe(a.scalars + SC_Y), beta = load_fe(a.scalars + SC_BETA), gamma = load_fe(a.scalars + SC_GAMMA);
    const uint32_t r_next = rot_row(idx, 1, a.rot_scale, mask), r_prev = rot_row(idx, -1, a.rot_scale, mask);
    const fe z = load_fe(a.product + idx), z_next = load_fe(a.product + r_next);
    const fe ai = load_fe(a.permuted_input + idx), ai_prev = load_fe(a.permuted_input + r_prev);
    const fe si = load_fe(a.permuted_table + idx);
    const fe l0 = load_fe(a.l0 + idx), l_last = load_fe(a.l_last + idx), l_active = load_fe(a.l_active + idx);
    const fe a_minus_s = Fr::sub(ai, si);
    fe v = prev;
    v = Fr::add(Fr::mul(v, y), Fr::mul(Fr::sub(Fr::one(), z), l0));
    v = Fr::add(Fr::mul(v, y), Fr::mul(Fr::sub(Fr::sqr(z), z), l_last));
    {
        const fe left = Fr::mul(Fr::mul(z_next, Fr::add(ai, beta)), Fr::add(si, gamma));
        const fe right = Fr::mul(z, value);
        v = Fr::add(Fr::mul(v, y), Fr::mul(Fr::sub(left, right), l_active));
    }
    v = Fr::add(Fr::mul(v, y), Fr::mul(a_minus_s, l0));
    v = Fr::add(Fr::mul(v, y), Fr::mul(Fr::mul(a_minus_s, Fr::sub(ai, ai_prev)), l_active));
    store_fe(a.out + idx, v);
}

struct PermArgs {
    fe* out;
    const fe* const* z;          // [nsets] permutation product cosets
    const fe* const* sigma;      // [ncols]
    const fe* const* colptr;     // [ncols] column cosets in permutation order
    uint32_t nsets, ncols, chunk;
    int32_t last_rot;
    const fe* l0;
    const fe* l_last;
    const fe* l_active;
    const fe* scalars;
    const fe* om_lo;             // extended_omega^i, i < 2^h
    const fe* om_hi;             // extended_omega^(i 2^h)
    uint32_t om_h;
    uint32_t log_size, rot_scale;
};

__global__ void __launch_bounds__(EV_THREADS) evalh_perm_kernel(const PermArgs a) {
    const uint32_t idx = blockIdx.x * EV_THREADS + threadIdx.x;
    const uint32_t mask = (1u << a.log_size) - 1;
    if (idx > mask) return;
    const fe y = load_fe(a.scalars + SC_Y), beta = load_fe(a.scalars + SC_BETA), gamma = load_fe(a.scalars + SC_GAMMA);
    const fe l0 = load_fe(a.l0 + idx), l_last = load_fe(a.l_last + idx), l_active = load_fe(a.l_active + idx);
    const uint32_t r_next = rot_row(idx, 1, a.rot_scale, mask), r_last = rot_row(idx, a.last_rot, a.rot_scale, mask);
    fe v = load_fe(a.out + idx);
    {
        const fe z0 = load_fe(a.z[0] + idx);
        v = Fr::add(Fr::mul(v, y), Fr::mul(Fr::sub(Fr::one(), z0), l0));
        const fe zl = load_fe(a.z[a.nsets - 1] + idx);
        v = Fr::add(Fr::mul(v, y), Fr::mul(Fr::sub(Fr::sqr(zl), zl), l_last));
    }
    for (uint32_t s = 1; s < a.nsets; s++) {
        const fe d = Fr::sub(load_fe(a.z[s] + idx), load_fe(a.z[s - 1] + r_last));
        v = Fr::add(Fr::mul(v, y), Fr::mul(d, l0));
    }
    // current_delta = beta * zeta * extended_omega^idx, then * DELTA per column (carried across sets)
    fe om = load_fe(a.om_lo + (idx & ((1u << a.om_h) - 1)));
    if (idx >> a.om_h) om = Fr::mul(om, load_fe(a.om_hi + (idx >> a.om_h)));
    fe zeta, delta;
#pragma unroll
    for (int i = 0; i < 8; i++) { zeta.l[i] = FrParams::ZETA[i]; delta.l[i] = FrParams::DELTA[i]; }
    fe current_delta = Fr::mul(Fr::mul(beta, zeta), om);
    for (uint32_t s = 0; s < a.nsets; s++) {
        const uint32_t c0 = s * a.chunk, c1 = c0 + a.chunk < a.ncols ? c0 + a.chunk : a.ncols;
        fe left = load_fe(a.z[s] + r_next), right = load_fe(a.z[s] + idx);
        for (uint32_t c = c0; c < c1; c++) {
            const fe val = load_fe(a.colptr[c] + idx);
            const fe sg = load_fe(a.sigma[c] + idx);
            left = Fr::mul(left, Fr::add(Fr::add(val, Fr::mul(beta, sg)), gamma));
            right = Fr::mul(right, Fr::add(Fr::add(val, current_delta), gamma));
            current_delta = Fr::mul(current_delta, delta);
        }
        v = Fr::add(Fr::mul(v, y), Fr::mul(Fr::sub(left, right), l_active));
    }
    store_fe(a.out + idx, v);
}

// ---- host-side lowering of a GraphEvaluator ---------------------------------------------------------
struct VS {
    uint32_t kind, a, b;
};
struct Calc {
    uint32_t op;
    VS src[2];
    VS start, factor;
    std::vector<VS> parts;
};
struct Program {
    std::vector<uint32_t> words;   // 4 per instruction
    uint32_t result = 0;
    uint32_t n_slots = 0;          // total slots (LDS + spill)
    uint32_t lds_slots = 1;        // slots kept in LDS
};

static bool parse_graph(const zkhip_graph* g, std::vector<Calc>& out, std::string& err) {
    const uint32_t* w = g->calc_words;
    const uint32_t* end = w + g->num_calc_words;
    auto need = [&](size_t k) { return (size_t)(end - w) >= k; };
    auto vs = [&](VS& v) {
        v.kind = w[0];
        v.a = w[1];
        v.b = w[2];
        w += 3;
    };
    out.resize(g->num_calculations);
    for (uint32_t t = 0; t < g->num_calculations; t++) {
        if (!need(1)) { err = "calc_words truncated"; return false; }
        Calc& c = out[t];
        c.op = *w++;
        switch (c.op) {
            case 0: case 1: case 2:
                if (!need(6)) { err = "calc_words truncated"; return false; }
                vs(c.src[0]);
                vs(c.src[1]);
                break;
            case 3: case 4: case 5: case 7:
                if (!need(3)) { err = "calc_words truncated"; return false; }
                vs(c.src[0]);
                break;
            case 6: {
                if (!need(7)) { err = "calc_words truncated"; return false; }
                vs(c.start);
                vs(c.factor);
                const uint32_t np = *w++;
                if (!need((size_t)np * 3)) { err = "calc_words truncated"; return false; }
                c.parts.resize(np);
                for (uint32_t p = 0; p < np; p++) vs(c.parts[p]);
                break;
            }
            default: err = "unknown calculation opcode"; return false;
        }
    }
    if (w != end) { err = "calc_words has trailing data"; return false; }
    return true;
}

struct Lowering {
    const zkhip_graph* g;
    const zkhip_evalh_desc* desc;
    std::vector<Calc> calcs;
    std::vector<uint32_t> val;      // operand word per calculation once emitted (virtual slot ids in K_SLOT payload)
    std::vector<uint8_t> done;
    std::vector<uint32_t> words;    // instruction stream with virtual slots
    uint32_t n_virtual = 0;
    uint32_t root = 0;
    std::string err;

    bool operand(const VS& v, uint32_t& word) {
        switch (v.kind) {
            case 0:
                if (v.a >= g->num_constants) { err = "constant index out of range"; return false; }
                word = (K_CONST << 28) | v.a;
                return true;
            case 1:
                if (v.a >= calcs.size() || !done[v.a]) { err = "intermediate used before definition"; return false; }
                word = val[v.a];
                return true;
            case 2: case 3: case 4: {
                uint32_t col;
                if (v.kind == 2) { if (v.a >= desc->num_fixed) { err = "fixed column out of range"; return false; } col = v.a; }
                else if (v.kind == 3) { if (v.a >= desc->num_advice) { err = "advice column out of range"; return false; } col = desc->num_fixed + v.a; }
                else { if (v.a >= desc->num_instance) { err = "instance column out of range"; return false; } col = desc->num_fixed + desc->num_advice + v.a; }
                if (v.b >= g->num_rotations) { err = "rotation index out of range"; return false; }
                if (col >= (1u << 16) || v.b >= (1u << 12)) { err = "too many columns or rotations"; return false; }
                word = (K_COL << 28) | (col << 12) | v.b;
                return true;
            }
            case 5:
                if (v.a >= desc->num_challenges) { err = "challenge index out of range"; return false; }
                word = (K_SCALAR << 28) | (SC_CHALLENGE0 + v.a);
                return true;
            case 6: word = (K_SCALAR << 28) | SC_BETA; return true;
            case 7: word = (K_SCALAR << 28) | SC_GAMMA; return true;
            case 8: word = (K_SCALAR << 28) | SC_THETA; return true;
            case 9: word = (K_SCALAR << 28) | SC_Y; return true;
            case 10: word = (K_PREV << 28); return true;
            default: err = "unknown ValueSource kind"; return false;
        }
    }
    void emit(uint32_t op, uint32_t dst, uint32_t a, uint32_t b = K_ZERO << 28, uint32_t c = K_ZERO << 28) {
        words.push_back(op | (dst << 8));
        words.push_back(a);
        words.push_back(b);
        words.push_back(c);
    }
    // iterative post-order: make calculation `t` available
    bool schedule(uint32_t t_root) {
        struct Frame { uint32_t t; uint32_t stage; };
        std::vector<Frame> st;
        st.push_back({t_root, 0});
        while (!st.empty()) {
            Frame& f = st.back();
            const uint32_t t = f.t;
            if (done[t]) { st.pop_back(); continue; }
            Calc& c = calcs[t];
            // collect intermediate dependencies in evaluation order
            auto dep = [&](const VS& v) -> bool {
                if (v.kind == 1 && v.a < calcs.size() && !done[v.a]) {
                    if (v.a >= t) { err = "calculation depends on a later one"; return true; }
                    st.push_back({v.a, 0});
                    return true;
                }
                return false;
            };
            if (c.op != 6) {
                const int ns = (c.op <= 2) ? 2 : 1;
                bool pushed = false;
                for (int k = 0; k < ns && !pushed; k++) pushed = dep(c.src[k]);
                if (!err.empty()) return false;
                if (pushed) continue;
                uint32_t a, b = K_ZERO << 28;
                if (!operand(c.src[0], a)) return false;
                if (ns == 2 && !operand(c.src[1], b)) return false;
                if (c.op == 7) {   // Store: alias the source
                    val[t] = a;
                } else {
                    const uint32_t dst = n_virtual++;
                    static const uint32_t map[6] = {D_ADD, D_SUB, D_MUL, D_SQR, D_DBL, D_NEG};
                    emit(map[c.op], dst, a, b);
                    val[t] = (K_SLOT << 28) | dst;
                }
                done[t] = 1;
                st.pop_back();
            } else {
                // Horner: stage 0 = start/factor deps, then one part at a time
                const bool is_root = (t == root);
                if (f.stage == 0) {
                    if (dep(c.start) || dep(c.factor)) { if (!err.empty()) return false; continue; }
                    uint32_t s;
                    if (!operand(c.start, s)) return false;
                    const uint32_t dst = is_root ? DST_ACC : n_virtual++;
                    c.src[0].kind = dst;   // remember the accumulator (reusing the unused src slot)
                    if (!(is_root && c.start.kind == 10)) emit(D_MOV, dst, s);   // the root accumulator starts as previous_value
                    f.stage = 1;
                }
                const uint32_t dst = c.src[0].kind;
                const uint32_t accw = (dst == DST_ACC) ? (K_ACC << 28) : ((K_SLOT << 28) | dst);
                bool pushed = false;
                while (f.stage - 1 < c.parts.size()) {
                    const VS& p = c.parts[f.stage - 1];
                    if (dep(p)) { pushed = true; break; }
                    uint32_t pw, fw;
                    if (!operand(p, pw) || !operand(c.factor, fw)) return false;
                    emit(D_MULADD, dst, accw, fw, pw);
                    st.back().stage++;   // `f` may dangle after push_back; index through st.back()
                }
                if (!err.empty()) return false;
                if (pushed) continue;
                val[t] = accw;
                done[t] = 1;
                st.pop_back();
            }
        }
        return true;
    }
};

// liveness-based slot assignment; the `lds_slots` slots with the smallest average gap between accesses stay on chip
// lds_budget: LDS slots the caller is willing to give one lane.  Returns through lds_slots_out the number actually
// used (the peak number of simultaneously live values when that is smaller), so the launch reserves no more LDS than
// the program needs and more workgroups fit on a CU.
static void assign_slots(std::vector<uint32_t>& words, uint32_t& result, uint32_t n_virtual, uint32_t lds_budget,
                         uint32_t& lds_slots_out, uint32_t& n_slots_out) {
    const size_t ni = words.size() / 4;
    std::vector<uint32_t> first(n_virtual, UINT32_MAX), last(n_virtual, 0), uses(n_virtual, 0);
    auto touch = [&](uint32_t v, uint32_t pc) {
        if (first[v] == UINT32_MAX) first[v] = pc;
        last[v] = std::max(last[v], pc);
        uses[v]++;
    };
    for (size_t pc = 0; pc < ni; pc++) {
        const uint32_t dst = words[4 * pc] >> 8;
        for (int k = 1; k <= 3; k++)
            if ((words[4 * pc + k] >> 28) == K_SLOT) touch(words[4 * pc + k] & 0x0fffffffu, (uint32_t)pc);
        if (dst != DST_ACC) touch(dst, (uint32_t)pc);
    }
    if ((result >> 28) == K_SLOT) touch(result & 0x0fffffffu, (uint32_t)ni);
    // gap metric; sort candidates for demotion by decreasing gap
    std::vector<uint32_t> order(n_virtual);
    for (uint32_t v = 0; v < n_virtual; v++) order[v] = v;
    auto gap = [&](uint32_t v) { return (double)(last[v] - first[v] + 1) / (double)std::max(1u, uses[v]); };
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return gap(x) > gap(y); });
    auto pressure = [&](const std::vector<uint8_t>& spilled) {
        std::vector<int32_t> diff(ni + 2, 0);
        for (uint32_t v = 0; v < n_virtual; v++)
            if (!spilled[v] && first[v] != UINT32_MAX) { diff[first[v]]++; diff[last[v] + 1]--; }
        int32_t cur = 0, mx = 0;
        for (size_t i = 0; i <= ni; i++) { cur += diff[i]; mx = std::max(mx, cur); }
        return (uint32_t)mx;
    };
    std::vector<uint8_t> spilled(n_virtual, 0);
    const uint32_t peak = pressure(spilled);
    const uint32_t lds_slots = std::max(1u, std::min(peak, lds_budget));
    lds_slots_out = lds_slots;
    if (peak > lds_slots) {
        // smallest prefix of `order` whose demotion brings the on-chip pressure within budget
        size_t lo = 0, hi = n_virtual;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            std::fill(spilled.begin(), spilled.end(), 0);
            for (size_t i = 0; i < mid; i++) spilled[order[i]] = 1;
            if (pressure(spilled) <= lds_slots) hi = mid;
            else lo = mid + 1;
        }
        std::fill(spilled.begin(), spilled.end(), 0);
        for (size_t i = 0; i < lo; i++) spilled[order[i]] = 1;
    }
    // linear scan within each class
    std::vector<uint32_t> phys(n_virtual, 0);
    std::vector<std::vector<uint32_t>> expire(ni + 2);
    for (uint32_t v = 0; v < n_virtual; v++)
        if (first[v] != UINT32_MAX) expire[last[v] + 1].push_back(v);
    std::vector<std::vector<uint32_t>> begin(ni + 2);
    for (uint32_t v = 0; v < n_virtual; v++)
        if (first[v] != UINT32_MAX) begin[first[v]].push_back(v);
    std::vector<uint32_t> free_lds, free_sp;
    uint32_t next_lds = 0, next_sp = 0;
    for (size_t pc = 0; pc <= ni; pc++) {
        for (uint32_t v : expire[pc]) (spilled[v] ? free_sp : free_lds).push_back(phys[v]);
        for (uint32_t v : begin[pc]) {
            if (spilled[v]) {
                if (!free_sp.empty()) { phys[v] = free_sp.back(); free_sp.pop_back(); }
                else phys[v] = lds_slots + next_sp++;
            } else {
                if (!free_lds.empty()) { phys[v] = free_lds.back(); free_lds.pop_back(); }
                else phys[v] = next_lds++;
            }
        }
    }
    auto remap = [&](uint32_t& w) {
        if ((w >> 28) == K_SLOT) w = (K_SLOT << 28) | phys[w & 0x0fffffffu];
    };
    for (size_t pc = 0; pc < ni; pc++) {
        const uint32_t dst = words[4 * pc] >> 8;
        if (dst != DST_ACC) words[4 * pc] = (words[4 * pc] & 0xff) | (phys[dst] << 8);
        for (int k = 1; k <= 3; k++) remap(words[4 * pc + k]);
    }
    remap(result);
    n_slots_out = lds_slots + next_sp;
}

static int lower_graph(zkhip_ctx* ctx, const zkhip_graph* g, const zkhip_evalh_desc* desc, uint32_t lds_budget, Program& out) {
    Lowering lw;
    lw.g = g;
    lw.desc = desc;
    if (!parse_graph(g, lw.calcs, lw.err)) return set_error(ctx, ZKHIP_ERR_INVALID, "evaluate_h: %s", lw.err.c_str());
    if (lw.calcs.empty()) {
        out.result = K_ZERO << 28;
        out.lds_slots = 1;
        return ZKHIP_OK;
    }
    lw.val.assign(lw.calcs.size(), 0);
    lw.done.assign(lw.calcs.size(), 0);
    lw.root = (uint32_t)lw.calcs.size() - 1;
    if (!lw.schedule(lw.root)) return set_error(ctx, ZKHIP_ERR_INVALID, "evaluate_h: %s", lw.err.c_str());
    out.words.swap(lw.words);
    out.result = lw.val[lw.root];
    ZK_CHECK(ctx, lw.n_virtual < DST_ACC, "evaluate_h: program too large");
    uint32_t lds_slots = 1;
    assign_slots(out.words, out.result, lw.n_virtual, lds_budget, lds_slots, out.n_slots);
    out.lds_slots = lds_slots;
    if (getenv("ZKHIP_DEBUG")) {
        size_t nmul = 0, ncol = 0, nspill = 0;
        for (size_t pc = 0; pc < out.words.size() / 4; pc++) {
            const uint32_t op = out.words[4 * pc] & 0xff;
            if (op == D_MUL || op == D_SQR || op == D_MULADD) nmul++;
            for (int k = 1; k <= 3; k++) {
                const uint32_t w = out.words[4 * pc + k];
                if ((w >> 28) == K_COL) ncol++;
                if ((w >> 28) == K_SLOT && (w & 0x0fffffffu) >= lds_slots) nspill++;
            }
            const uint32_t dst = out.words[4 * pc] >> 8;
            if (dst != DST_ACC && dst >= lds_slots) nspill++;
        }
        fprintf(stderr, "[zkhip] graph: %u calculations -> %zu instructions (%zu multiplications, %zu column reads, %zu spill accesses), %u virtual values, %u slots (%u in LDS)\n",
                g->num_calculations, out.words.size() / 4, nmul, ncol, nspill, lw.n_virtual, out.n_slots, lds_slots);
    }
    return ZKHIP_OK;
}

// ---- device buffer bookkeeping for one call -----------------------------------------------------------
struct Arena {
    zkhip_ctx* ctx;
    std::vector<void*> owned;
    ~Arena() {
        for (void* p : owned) (void)hipFree(p);
    }
    int alloc(size_t bytes, void** out) {
        ZK_HIP(ctx, hipMalloc(out, bytes ? bytes : 1));
        owned.push_back(*out);
        return ZKHIP_OK;
    }
    template <class T>
    int upload(const T* host, size_t count, T** out) {
        void* d;
        int rc = alloc(count * sizeof(T), &d);
        if (rc) return rc;
        ZK_HIP(ctx, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        *out = (T*)d;
        return ZKHIP_OK;
    }
};

}  // namespace zkhip

extern "C" int zkhip_evaluate_h(zkhip_domain* dom, const zkhip_evalh_desc* desc, uint64_t* out) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, desc && out, "evaluate_h: null descriptor or output");
    ZK_CHECK(ctx, desc->mem <= 1 && desc->poly_form <= 1, "evaluate_h: bad mem / poly_form");
    ZK_CHECK(ctx, desc->l0 && desc->l_last && desc->l_active_row, "evaluate_h: l0 / l_last / l_active_row missing");
    ZK_CHECK(ctx, !desc->num_fixed || desc->fixed_cosets, "evaluate_h: fixed_cosets missing");
    ZK_CHECK(ctx, !desc->num_advice || desc->advice, "evaluate_h: advice missing");
    ZK_CHECK(ctx, !desc->num_instance || desc->instance, "evaluate_h: instance missing");
    ZK_CHECK(ctx, !desc->num_challenges || desc->challenges, "evaluate_h: challenges missing");
    ZK_CHECK(ctx, !desc->num_lookups || desc->lookups, "evaluate_h: lookups missing");
    if (desc->num_perm_sets) {
        ZK_CHECK(ctx, desc->perm_chunk_len >= 1, "evaluate_h: perm_chunk_len = 0");
        ZK_CHECK(ctx, desc->perm_column_kind && desc->perm_column_index && desc->perm_sigma_cosets && desc->perm_product_cosets,
                 "evaluate_h: permutation arrays missing");
        ZK_CHECK(ctx, (desc->num_perm_columns + desc->perm_chunk_len - 1) / desc->perm_chunk_len == desc->num_perm_sets,
                 "evaluate_h: %u permutation columns in chunks of %u do not make %u sets", desc->num_perm_columns,
                 desc->perm_chunk_len, desc->num_perm_sets);
    }
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t ek = dom->extended_k, k = dom->k;
    const size_t size = (size_t)1 << ek, n = (size_t)1 << k;
    const bool host = desc->mem == ZKHIP_MEM_HOST;
    const bool coeff = desc->poly_form == ZKHIP_FORM_COEFF;
    Arena ar{ctx, {}};
    int rc;

    // Bring one polynomial to an extended coset on the device.  `ext_in`: already extended.
    fe *tmp0 = nullptr, *tmp1 = nullptr;
    if (coeff) {
        rc = ensure_scratch(ctx, 2 * size * 32);
        if (rc) return rc;
        tmp0 = (fe*)ctx->scratch;
        tmp1 = tmp0 + size;
    }
    auto resolve = [&](const uint64_t* p, bool ext_in, const fe** outp) -> int {
        ZK_CHECK(ctx, p != nullptr, "evaluate_h: null polynomial pointer");
        if (ext_in) {
            if (!host) { *outp = (const fe*)p; return ZKHIP_OK; }
            fe* d;
            int r = ar.upload((const fe*)p, size, &d);
            if (r) return r;
            *outp = d;
            return ZKHIP_OK;
        }
        const fe* d_coeff = (const fe*)p;
        if (host) {
            fe* d;
            int r = ar.upload((const fe*)p, n, &d);
            if (r) return r;
            d_coeff = d;
        }
        void* d_ext;
        int r = ar.alloc(size * 32, &d_ext);
        if (r) return r;
        r = domain_coeff_to_extended(dom, d_coeff, (fe*)d_ext, 1, tmp0, tmp1);
        if (r) return r;
        *outp = (const fe*)d_ext;
        return ZKHIP_OK;
    };

    // unified column table: fixed | advice | instance
    const uint32_t ncols = desc->num_fixed + desc->num_advice + desc->num_instance;
    std::vector<const fe*> cols(ncols ? ncols : 1, nullptr);
    for (uint32_t i = 0; i < desc->num_fixed; i++)
        if ((rc = resolve(desc->fixed_cosets[i], true, &cols[i]))) return rc;
    for (uint32_t i = 0; i < desc->num_advice; i++)
        if ((rc = resolve(desc->advice[i], !coeff, &cols[desc->num_fixed + i]))) return rc;
    for (uint32_t i = 0; i < desc->num_instance; i++)
        if ((rc = resolve(desc->instance[i], !coeff, &cols[desc->num_fixed + desc->num_advice + i]))) return rc;
    const fe** d_cols;
    if ((rc = ar.upload(cols.data(), cols.size(), &d_cols))) return rc;
    const fe *d_l0, *d_l_last, *d_l_active;
    if ((rc = resolve(desc->l0, true, &d_l0))) return rc;
    if ((rc = resolve(desc->l_last, true, &d_l_last))) return rc;
    if ((rc = resolve(desc->l_active_row, true, &d_l_active))) return rc;

    // scalar table
    std::vector<uint64_t> scal((SC_CHALLENGE0 + desc->num_challenges) * 4);
    memcpy(&scal[4 * SC_BETA], desc->beta, 32);
    memcpy(&scal[4 * SC_GAMMA], desc->gamma, 32);
    memcpy(&scal[4 * SC_THETA], desc->theta, 32);
    memcpy(&scal[4 * SC_Y], desc->y, 32);
    if (desc->num_challenges) memcpy(&scal[4 * SC_CHALLENGE0], desc->challenges, (size_t)desc->num_challenges * 32);
    fe* d_scal;
    if ((rc = ar.upload((const fe*)scal.data(), scal.size() / 4, &d_scal))) return rc;

    // output buffer
    fe* d_out;
    if (host) {
        void* p;
        if ((rc = ar.alloc(size * 32, &p))) return rc;
        d_out = (fe*)p;
    } else {
        d_out = (fe*)out;
    }
    ZK_HIP(ctx, hipMemsetAsync(d_out, 0, size * 32, ctx->stream));

    // LDS budget per lane: 6 slots of 32 B keep three 256-lane workgroups on a CU (the register file allows three)
    const char* env_slots = getenv("ZKHIP_EVALH_LDS_SLOTS");
    const uint32_t lds_budget = env_slots && *env_slots ? (uint32_t)atoi(env_slots) : 6;
    const dim3 grid((unsigned)((size + EV_THREADS - 1) / EV_THREADS));
    ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(evalh_interp_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    auto run_graph = [&](const zkhip_graph* g, uint32_t mode, const fe* prod, const fe* pin, const fe* ptab,
                         const char* family) -> int {
        ZK_CHECK(ctx, (g->constants || !g->num_constants) && (g->rotations || !g->num_rotations) &&
                          (g->calc_words || !g->num_calc_words),
                 "evaluate_h: graph arrays missing");
        Program prog;
        int r = lower_graph(ctx, g, desc, lds_budget, prog);
        if (r) return r;
        const uint32_t lds_slots = prog.lds_slots;
        const size_t lds_bytes = (size_t)lds_slots * EV_THREADS * sizeof(fe);
        InterpArgs a;
        memset(&a, 0, sizeof a);
        uint32_t* d_prog;
        if ((r = ar.upload(prog.words.data(), prog.words.size(), &d_prog))) return r;
        fe* d_consts;
        if ((r = ar.upload((const fe*)g->constants, g->num_constants, &d_consts))) return r;
        int32_t* d_rots;
        if ((r = ar.upload(g->rotations, g->num_rotations, &d_rots))) return r;
        a.prog = d_prog;
        a.n_instr = (uint32_t)(prog.words.size() / 4);
        a.result = prog.result;
        a.consts = d_consts;
        a.scalars = d_scal;
        a.cols = d_cols;
        a.rots = d_rots;
        a.lds_slots = lds_slots;
        if (prog.n_slots > lds_slots) {
            void* sp;
            if ((r = ar.alloc((size_t)(prog.n_slots - lds_slots) * size * 32, &sp))) return r;
            a.spill = (fe*)sp;
        }
        a.log_size = ek;
        a.rot_scale = 1u << (ek - k);
        a.out = d_out;
        a.mode = mode;
        a.product = prod;
        a.permuted_input = pin;
        a.permuted_table = ptab;
        a.l0 = d_l0;
        a.l_last = d_l_last;
        a.l_active = d_l_active;
        TimedRegion tr(ctx, family);
        hipLaunchKernelGGL(evalh_interp_kernel, grid, dim3(EV_THREADS), lds_bytes, ctx->stream, a);
        ZK_HIP(ctx, hipGetLastError());
        return ZKHIP_OK;
    };

    // 1. custom gates
    if (desc->custom_gates) {
        rc = run_graph(desc->custom_gates, 0, nullptr, nullptr, nullptr, "evalh_gates");
        if (rc) return rc;
    }

    // 2. permutation argument
    if (desc->num_perm_sets) {
        std::vector<const fe*> z(desc->num_perm_sets), sigma(desc->num_perm_columns), colp(desc->num_perm_columns);
        for (uint32_t s = 0; s < desc->num_perm_sets; s++)
            if ((rc = resolve(desc->perm_product_cosets[s], true, &z[s]))) return rc;
        for (uint32_t c = 0; c < desc->num_perm_columns; c++) {
            if ((rc = resolve(desc->perm_sigma_cosets[c], true, &sigma[c]))) return rc;
            const uint32_t kind = desc->perm_column_kind[c], ci = desc->perm_column_index[c];
            uint32_t u;
            if (kind == 2) { ZK_CHECK(ctx, ci < desc->num_fixed, "evaluate_h: permutation fixed column out of range"); u = ci; }
            else if (kind == 3) { ZK_CHECK(ctx, ci < desc->num_advice, "evaluate_h: permutation advice column out of range"); u = desc->num_fixed + ci; }
            else if (kind == 4) { ZK_CHECK(ctx, ci < desc->num_instance, "evaluate_h: permutation instance column out of range"); u = desc->num_fixed + desc->num_advice + ci; }
            else return set_error(ctx, ZKHIP_ERR_INVALID, "evaluate_h: permutation column kind %u", kind);
            colp[c] = cols[u];
        }
        PermArgs pa;
        memset(&pa, 0, sizeof pa);
        const fe **d_z, **d_sigma, **d_colp;
        if ((rc = ar.upload(z.data(), z.size(), &d_z))) return rc;
        if ((rc = ar.upload(sigma.data(), sigma.size(), &d_sigma))) return rc;
        if ((rc = ar.upload(colp.data(), colp.size(), &d_colp))) return rc;
        pa.out = d_out;
        pa.z = d_z;
        pa.sigma = d_sigma;
        pa.colptr = d_colp;
        pa.nsets = desc->num_perm_sets;
        pa.ncols = desc->num_perm_columns;
        pa.chunk = desc->perm_chunk_len;
        pa.last_rot = -((int32_t)desc->blinding_factors + 1);
        pa.l0 = d_l0;
        pa.l_last = d_l_last;
        pa.l_active = d_l_active;
        pa.scalars = d_scal;
        pa.om_lo = (const fe*)dom->plan_ext_fwd->d_lo;
        pa.om_hi = (const fe*)dom->plan_ext_fwd->d_hi;
        pa.om_h = dom->plan_ext_fwd->h;
        pa.log_size = ek;
        pa.rot_scale = 1u << (ek - k);
        TimedRegion tr(ctx, "evalh_perm");
        hipLaunchKernelGGL(evalh_perm_kernel, grid, dim3(EV_THREADS), 0, ctx->stream, pa);
        ZK_HIP(ctx, hipGetLastError());
    }

    // 3. lookups, one at a time (their three cosets only live for the duration of their pass)
    for (uint32_t i = 0; i < desc->num_lookups; i++) {
        const zkhip_lookup& lk = desc->lookups[i];
        const fe *prod, *pin, *ptab;
        if ((rc = resolve(lk.product, !coeff, &prod))) return rc;
        if ((rc = resolve(lk.permuted_input, !coeff, &pin))) return rc;
        if ((rc = resolve(lk.permuted_table, !coeff, &ptab))) return rc;
        rc = run_graph(&lk.graph, 1, prod, pin, ptab, "evalh_lookup");
        if (rc) return rc;
    }

    if (host) ZK_HIP(ctx, hipMemcpyAsync(out, d_out, size * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}
