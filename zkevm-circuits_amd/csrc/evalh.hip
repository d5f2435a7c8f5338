// `plonk::evaluation::Evaluator::evaluate_h` on gfx950: quotient numerator over the extended coset.
//
// One lane per extended row, ONE kernel launch for the whole function.  The host fuses custom gates, the
// permutation argument and every lookup argument into a single register-machine program:
//   * each serialised `GraphEvaluator` is lowered just-in-time: only calculations reachable from its result are
//     emitted, each right before its first use; `Store` of a column / challenge is never materialised (users read the
//     source); `Horner(start, parts, factor)` becomes one multiply-add per part;
//   * the running value `values[idx] = values[idx] * y + term` of upstream's row loops lives in a register
//     accumulator for the whole program, so the output is written once instead of once per argument;
//   * permutation and lookup terms are emitted as instructions following upstream's formulas; their cosets, l0,
//     l_last, l_active_row and the row's point zeta * extended_omega^idx join the unified column table;
//   * values get slots by liveness; the slots with the densest use stay on chip (slot 0 in registers, the next ones
//     in LDS as [slot][digit][lane]), the rest spill to a [slot][digit][row] buffer in HBM.  The launch reserves only
//     the LDS the program needs, so that four 256-lane workgroups share a CU;
//   * arithmetic runs in the lazy signed-digit form of wide.hip.h (nine 29-bit digits, Montgomery radix 2^261): a
//     multiply-add is one Montgomery product, nine digit additions and a carry pass, with no comparison against p;
//     the host bounds every value's magnitude while it builds the program and the kernel canonicalises the result.
// Every instruction is r = (a * b | a) (+ c | - c | nothing), r = c - a * b, or r = a * b +- c * d where both products
// share one Montgomery reduction; a peephole pass over the finished program moves a product that feeds a single
// addition or subtraction into its consumer.  The program is wave-uniform: its words arrive by scalar loads with
// column pointers and row offsets already resolved, every branch is a scalar branch, and the column / constant reads
// of instruction pc + 1 are issued as soon as instruction pc has computed; the four waves per SIMD hide their latency
// (holding a second set of staged operands across the products costs a wave per SIMD and more than it gains).
// Column reads are (row + rotation * 2^(ext_k - k)) mod 2^ext_k, i.e. 32 contiguous bytes per lane.
// Results equal upstream's bit for bit: field arithmetic is exact, only the order of independent operations differs.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include "domain.h"
#include "wide.hip.h"

using namespace zkhip;

namespace zkhip {

constexpr int EV_THREADS = 256;

// instruction word 0: bit 0 = multiply a * b, bit 1 = add c, bit 2 = subtract c, bit 3 = carry-normalise the result,
// bits 4 / 5 = carry-normalise operand a / b after fetching it, bit 6 = reverse subtraction (c - a * b),
// bit 7 = second product: a * b +- c * d under one Montgomery reduction, bits 8 / 9 = carry-normalise operand c / d
// of a second product, bits 12.. = destination
constexpr uint32_t F_MUL = 1, F_ADD = 2, F_SUB = 4, F_NORM = 8, F_NORM_A = 16, F_NORM_B = 32, F_RSUB = 64, F_MUL2 = 128,
                   F_NORM_C = 256, F_NORM_D = 512,
                   F_PRE = 1024,       // with F_MUL alone: r = a (b + c), the sum formed on the way into the multiplier
                   F_PRESUB = 2048;    // with F_PRE: r = a (b - c)
constexpr uint32_t F_OPS = F_MUL | F_ADD | F_SUB | F_RSUB | F_MUL2 | F_PRE | F_PRESUB;
constexpr uint32_t F_NORMS = F_NORM | F_NORM_A | F_NORM_B | F_NORM_C | F_NORM_D;
constexpr uint32_t DST_SHIFT = 12;
enum DevKind : uint32_t { K_CONST = 0, K_SLOT, K_COL, K_SCALAR, K_ACC, K_ZERO };
constexpr uint32_t DST_ACC = 0xfffffu;         // in a linked instruction
constexpr uint32_t DST_ACC_V = 0xffffffffu;    // in the program under construction (virtual destinations)
constexpr uint32_t W_ZERO = K_ZERO << 28, W_ACC = K_ACC << 28;

// scalar table layout
constexpr uint32_t SC_BETA = 0, SC_GAMMA = 1, SC_THETA = 2, SC_Y = 3, SC_ONE = 4, SC_CHALLENGE0 = 5;

// Linked program: 16 words per instruction (13 used),
//   [0] flags | dst << 12, then for each operand a, b, c, d: header = kind << 28 | payload, pointer low, pointer high.
//   L_ROW : pointer = column base, payload = signed row offset of the rotation (28 bits)
//   L_UNI : pointer = one field element read by every lane (constant / scalar)
//   L_SLOT: payload = value slot;  L_ACC: the running value;  L_ZERO
// The words are read through the constant address space, i.e. by scalar loads: decode never touches the vector
// memory path and needs no table lookups.
enum LinkKind : uint32_t { L_ZERO = 0, L_ROW, L_UNI, L_SLOT, L_ACC };
constexpr uint32_t INS_WORDS = 16;
typedef const uint32_t __attribute__((address_space(4))) * prog_ptr;

struct Opnd {
    uint32_t hdr;
    const fe* ptr;
};
// One program of a launch that runs several (blockIdx.y selects it): where its linked words start, and the fields of
// InterpArgs that differ from program to program.  Read through the constant address space, like the program itself.
struct ProgEntry {
    uint32_t word_offset, n_instr, result_hdr, lds_slots;
    fe* out;
    uint64_t pad;
};
typedef const ProgEntry __attribute__((address_space(4))) * entry_ptr;
struct InterpArgs {
    const ProgEntry* table;     // null: one program, described by the fields below
    uint32_t n_instr;
    uint32_t result_hdr;        // operand of the final value (slot / running value / zero)
    uint32_t lds_slots;
    uint32_t log_size;          // log2 of the rows of every column (rotations wrap around at this size)
    uint32_t row_first, rows;   // the rows evaluated: [row_first, row_first + rows); out and the spill buffer hold `rows` rows
    int32_t* spill;             // [slot - lds_slots][digit][rows]
    fe* out;
};

__device__ __forceinline__ Opnd ev_operand(prog_ptr w) {
    Opnd o;
    o.hdr = w[0];
    o.ptr = reinterpret_cast<const fe*>(((uint64_t)w[2] << 32) | w[1]);
    return o;
}
// Issues the memory read of a column / constant operand, through the global address space (a pointer rebuilt from
// program words would otherwise be read with flat loads, which also count against the LDS / scalar-load waits).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) * global_u4;
__device__ __forceinline__ fe ev_raw(const Opnd& o, uint32_t idx, uint32_t mask) {
    const uint32_t kind = o.hdr >> 28;
    const int32_t delta = (int32_t)(o.hdr << 4) >> 4;
    const uint32_t row = kind == L_ROW ? ((uint32_t)((int32_t)idx + delta) & mask) : 0u;
    const global_u4 q = (global_u4)(uintptr_t)(o.ptr + row);
    const u32x4 a = q[0], b = q[1];
    fe v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}
// Values are held in the lazy signed-digit form of wide.hip.h, 2^261 Montgomery domain, always normalised.  Columns
// enter through from_fe_shl5 (value up to 32 p); the host tracks magnitudes and keeps every stored value below EV_LIMIT p.
// Value slots: slot 0 is a register of the lane, slots 1 .. lds_slots-1 live in LDS, the rest in the spill buffer.
__device__ __forceinline__ fw ev_value(const InterpArgs& a, const Opnd& o, const fe& raw, const int32_t* lds, const fw& acc,
                                       const fw& reg0, uint32_t local) {
    const uint32_t kind = o.hdr >> 28, pl = o.hdr & 0x0fffffffu;
    switch (kind) {
        case L_ROW: return FrW::from_fe_shl5(raw);
        case L_UNI: return FrW::from_fe(raw);
        case L_SLOT: {
            if (pl == 0) return reg0;
            fw v;
            if (pl < a.lds_slots) {
#pragma unroll
                for (int d = 0; d < 9; d++) v.d[d] = lds[((pl - 1) * 9 + d) * EV_THREADS + threadIdx.x];
            } else {
                const int32_t* base = a.spill + (size_t)(pl - a.lds_slots) * 9 * a.rows + local;
#pragma unroll
                for (int d = 0; d < 9; d++) v.d[d] = base[(size_t)d * a.rows];
            }
            return v;
        }
        case L_ACC: return acc;
        default: return FrW::zero();
    }
}

// decoded instruction plus its raw memory operands, loads in flight
// An instruction reads at most three operands from memory: with a second product (F_MUL2) operand a is a value slot or
// the running value, and the staging register `rad` holds d's memory operand instead of a's.
struct Staged {
    uint32_t w0;
    Opnd a, b, c, d;
    fe rad, rb, rc;
};
__device__ __forceinline__ void ev_stage(Staged& s, prog_ptr w, uint32_t idx, uint32_t mask) {
    s.w0 = w[0];
    s.a = ev_operand(w + 1);
    s.b = ev_operand(w + 4);
    s.c = ev_operand(w + 7);
    s.d = ev_operand(w + 10);
    const Opnd& ad = (s.w0 & F_MUL2) ? s.d : s.a;
    const uint32_t kad = ad.hdr >> 28, kb = s.b.hdr >> 28, kc = s.c.hdr >> 28;
    if (kad == L_ROW || kad == L_UNI) s.rad = ev_raw(ad, idx, mask);
    if (kb == L_ROW || kb == L_UNI) s.rb = ev_raw(s.b, idx, mask);
    if (kc == L_ROW || kc == L_UNI) s.rc = ev_raw(s.c, idx, mask);
}
// the multiplier of the interpreter: one dependent chain of multiply-adds per column (see Wide::chain)
#ifndef ZKHIP_EVALH_CHAIN_MUL
#define ZKHIP_EVALH_CHAIN_MUL 1
#endif
using EvW = Wide<FrParams, ZKHIP_EVALH_CHAIN_MUL != 0>;
__device__ __forceinline__ void ev_execute(const InterpArgs& a, const Staged& s, int32_t* lds, fw& acc, fw& reg0, uint32_t idx) {
    // idx: the lane's row inside the evaluated window (value slots are per lane; columns were read while staging)
    const uint32_t w0 = s.w0, dst = w0 >> DST_SHIFT;
    fw t;
    if (w0 & F_MUL2) {
        // a b +- c d: the column-wise multiplier adds both products' terms to one running accumulator and retires them with
        // one reduction (a is a slot or the running value, never memory)
        fw c = ev_value(a, s.c, s.rc, lds, acc, reg0, idx);
        if (w0 & F_NORM_C) c = FrW::norm(c);
        fw d = ev_value(a, s.d, s.rad, lds, acc, reg0, idx);
        if (w0 & F_NORM_D) d = FrW::norm(d);
        t = ev_value(a, s.a, Fr::zero(), lds, acc, reg0, idx);
        if (w0 & F_NORM_A) t = FrW::norm(t);
        fw u = ev_value(a, s.b, s.rb, lds, acc, reg0, idx);
        if (w0 & F_NORM_B) u = FrW::norm(u);
        t = EvW::mul_add_mul(t, u, FrW::cneg(c, (w0 & F_SUB) != 0), d);
    } else {
        t = ev_value(a, s.a, s.rad, lds, acc, reg0, idx);
        if (w0 & F_NORM_A) t = FrW::norm(t);
        if (w0 & F_MUL) {
            fw u = ev_value(a, s.b, s.rb, lds, acc, reg0, idx);
            if (w0 & F_PRE) {
                const fw c = ev_value(a, s.c, s.rc, lds, acc, reg0, idx);
                u = (w0 & F_PRESUB) ? FrW::sub(u, c) : FrW::add(u, c);
            }
            if (w0 & F_NORM_B) u = FrW::norm(u);
            t = EvW::mul(t, u);
        }
        if (w0 & (F_ADD | F_SUB | F_RSUB)) {
            const fw c = ev_value(a, s.c, s.rc, lds, acc, reg0, idx);
            t = (w0 & F_ADD) ? FrW::add(t, c) : FrW::sub(t, c);
            if (w0 & F_RSUB) t = FrW::neg(t);
            if (w0 & F_NORM) t = FrW::norm(t);
        }
    }
    if (dst == DST_ACC) {
        acc = t;
    } else if (dst == 0) {
        reg0 = t;
    } else if (dst < a.lds_slots) {
#pragma unroll
        for (int d = 0; d < 9; d++) lds[((dst - 1) * 9 + d) * EV_THREADS + threadIdx.x] = t.d[d];
    } else {
        int32_t* base = a.spill + (size_t)(dst - a.lds_slots) * 9 * a.rows + idx;
#pragma unroll
        for (int d = 0; d < 9; d++) base[(size_t)d * a.rows] = t.d[d];
    }
}

__global__ void __launch_bounds__(EV_THREADS) evalh_interp_kernel(const uint32_t* prog_generic, const InterpArgs a_in) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* lds = reinterpret_cast<int32_t*>(smem);
    prog_ptr prog = (prog_ptr)prog_generic;
    InterpArgs a = a_in;
    if (a_in.table) {
        const entry_ptr e = (entry_ptr)a_in.table + blockIdx.y;
        prog += e->word_offset;
        a.n_instr = e->n_instr;
        a.result_hdr = e->result_hdr;
        a.lds_slots = e->lds_slots;
        a.out = e->out;
    }
    const uint32_t local = blockIdx.x * EV_THREADS + threadIdx.x;   // row inside the evaluated window
    const uint32_t mask = (1u << a.log_size) - 1;
    if (local >= a.rows) return;   // windows that are no multiple of a workgroup
    const uint32_t idx = a.row_first + local;                       // row of the columns
    fw acc = FrW::zero(), reg0 = FrW::zero();
    // instruction pc + 1 is decoded and its column / constant loads are issued right after instruction pc has computed
    Staged st;
    if (a.n_instr) ev_stage(st, prog, idx, mask);
    for (uint32_t pc = 0; pc < a.n_instr; pc++) {
        ev_execute(a, st, lds, acc, reg0, local);
        if (pc + 1 < a.n_instr) ev_stage(st, prog + (size_t)(pc + 1) * INS_WORDS, idx, mask);
    }
    // back to the 2^256 domain, canonical
    Opnd ro;
    ro.hdr = a.result_hdr;
    ro.ptr = nullptr;
    const fw r = ev_value(a, ro, Fr::zero(), lds, acc, reg0, local);
    store_fe(a.out + local, FrW::to_fe(FrW::mul(r, FrW::r256())));
}

// x[idx] = zeta * extended_omega^idx: the point of extended row idx
__global__ void __launch_bounds__(256) xpoints_kernel(fe* x, const fe* om_lo, const fe* om_hi, uint32_t h, uint32_t size) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= size) return;
    fe zeta;
#pragma unroll
    for (int i = 0; i < 8; i++) zeta.l[i] = FrParams::ZETA[i];
    fe om = load_fe(om_lo + (idx & ((1u << h) - 1)));
    if (idx >> h) om = Fr::mul(om, load_fe(om_hi + (idx >> h)));
    store_fe(x + idx, Fr::mul(om, zeta));
}

// ---- host-side program construction ------------------------------------------------------------------
struct VS {
    uint32_t kind, a, b;
};
struct Calc {
    uint32_t op;
    VS src[2];
    VS start, factor;
    std::vector<VS> parts;
    uint32_t acc_dst = 0;
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

// The fused program under construction: six words per instruction (flags, virtual destination, operands a b c d), the
// merged constant and rotation tables.
constexpr size_t IW = 6;
struct Builder {
    std::vector<uint32_t> words;
    std::vector<uint64_t> consts;      // 4 limbs each
    std::vector<int32_t> rots;
    uint32_t n_virtual = 0;

    uint32_t rot(int32_t r) {
        for (size_t i = 0; i < rots.size(); i++)
            if (rots[i] == r) return (uint32_t)i;
        rots.push_back(r);
        return (uint32_t)rots.size() - 1;
    }
    uint32_t col(uint32_t c, int32_t r) { return (K_COL << 28) | (c << 12) | rot(r); }
    static uint32_t scalar(uint32_t i) { return (K_SCALAR << 28) | i; }
    static uint32_t slot(uint32_t v) { return (K_SLOT << 28) | v; }
    uint32_t fresh() {
        vbound.push_back(0.0);
        vdig.push_back(Digits());
        return n_virtual++;
    }
    // Upper bounds on |value| / p.  A multiplication maps (x, y) to below x y / 147 + 1 (Montgomery radix 2^261 over
    // p < 2^254); a column read is below 32; additions add.  Every stored value is kept below EV_LIMIT so that its top
    // digit stays a valid multiplier operand: a value that would exceed it is multiplied by one right away.
    static constexpr double EV_LIMIT = 120.0, EV_CONTRACT = 147.0;
    std::vector<double> vbound;
    double acc_bound = 0.0;
    double bound(uint32_t w) const {
        switch (w >> 28) {
            case K_CONST: case K_SCALAR: return 1.0;
            case K_COL: return 32.0;
            case K_SLOT: return vbound[w & 0x0fffffffu];
            case K_ACC: return acc_bound;
            default: return 0.0;
        }
    }
    // Digit bounds of every value, in units of 2^29 (digits 0..7; the top digit is covered by the value bound above).
    // Fetched columns / constants and products are normalised: [0, 1].  A product needs |a_i| |b_j| <= 2 * 2^58 so that
    // the 64-bit columns hold; sums must stay inside int32.  The carry pass after an addition is only emitted when
    // the result would leave [-2, 2]; an operand is normalised at fetch when a product would otherwise overflow.
    struct Digits {
        double lo = 0.0, hi = 0.0;
        double mag() const { return std::max(-lo, hi); }
    };
    std::vector<Digits> vdig;
    Digits acc_dig;
    Digits digits(uint32_t w) const {
        switch (w >> 28) {
            case K_CONST: case K_SCALAR: case K_COL: return {0.0, 1.0};
            case K_SLOT: return vdig[w & 0x0fffffffu];
            case K_ACC: return acc_dig;
            default: return {0.0, 0.0};
        }
    }
    // Magnitude bookkeeping of one instruction: sets the carry-normalisation flags it needs, records the bounds of its
    // result and returns the flags; `v_out` is the value bound of the result.
    uint32_t track(uint32_t flags, uint32_t dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d, double& v_out) {
        flags &= ~F_NORMS;
        double v = bound(a);
        Digits t = digits(a);
        if (flags & F_MUL2) {
            // both products share the columns: |a_i| |b_j| + |c_i| |d_j| <= 2 * 2^58; widest operand first
            double m[4] = {t.mag(), digits(b).mag(), digits(c).mag(), digits(d).mag()};
            const uint32_t nf[4] = {F_NORM_A, F_NORM_B, F_NORM_C, F_NORM_D};
            while (m[0] * m[1] + m[2] * m[3] > 2.0) {
                int w = 0;
                for (int q = 1; q < 4; q++)
                    if (m[q] > m[w]) w = q;
                if (m[w] <= 1.0) break;   // all normalised: 1 + 1
                flags |= nf[w];
                m[w] = 1.0;
            }
            v = (v * bound(b) + bound(c) * bound(d)) / EV_CONTRACT + 1.0;
            t = {0.0, 1.0};
        } else {
            if (flags & F_MUL) {
                double vb = bound(b);
                Digits u = digits(b);
                if (flags & F_PRE) {                       // the second factor is b +- c (both read from memory: digits in [0, 1])
                    const Digits w = digits(c);
                    vb += bound(c);
                    u = (flags & F_PRESUB) ? Digits{u.lo - w.hi, u.hi - w.lo} : Digits{u.lo + w.lo, u.hi + w.hi};
                }
                v = v * vb / EV_CONTRACT + 1.0;
                if (t.mag() * u.mag() > 2.0 && t.mag() > 1.0) {
                    flags |= F_NORM_A;
                    t = {0.0, 1.0};
                }
                if (t.mag() * u.mag() > 2.0) flags |= F_NORM_B;
                t = {0.0, 1.0};
            }
            if (!(flags & F_PRE) && (flags & (F_ADD | F_SUB | F_RSUB))) {
                v += bound(c);
                const Digits w = digits(c);
                if (!(flags & F_MUL) && t.mag() + w.mag() > 3.5) {   // int32 headroom of the digit sums
                    flags |= F_NORM_A;
                    t = {0.0, 1.0};
                }
                if (flags & F_ADD) t = {t.lo + w.lo, t.hi + w.hi};
                else if (flags & F_SUB) t = {t.lo - w.hi, t.hi - w.lo};
                else t = {w.lo - t.hi, w.hi - t.lo};
                if (t.mag() > 2.0) {
                    flags |= F_NORM;
                    t = {0.0, 1.0};
                }
            }
        }
        (dst == DST_ACC_V ? acc_bound : vbound[dst]) = v;
        (dst == DST_ACC_V ? acc_dig : vdig[dst]) = t;
        v_out = v;
        return flags;
    }
    void emit(uint32_t flags, uint32_t dst, uint32_t a, uint32_t b = W_ZERO, uint32_t c = W_ZERO, uint32_t d = W_ZERO) {
        double v;
        flags = track(flags, dst, a, b, c, d, v);
        const uint32_t ins[IW] = {flags, dst, a, b, c, d};
        words.insert(words.end(), ins, ins + IW);
        // renormalise the value through a multiplication by one
        if (v > EV_LIMIT) emit(F_MUL, dst, dst == DST_ACC_V ? W_ACC : slot(dst), scalar(SC_ONE));
    }

    // Peephole pass over the finished program.  A value that one plain multiplication p * q produces for a single
    // consumer, itself an addition or subtraction, moves into the consumer:
    //     x + p q, p q +- x  ->  one multiply-add;      x - p q  ->  reverse multiply-subtract;
    //     a b +- p q         ->  a second product in the same instruction, retired by the one Montgomery reduction.
    // The operands of the moved product must not change in between (values written once; never the running value).
    // Bounds are then recomputed from scratch: a fused result is never larger than what it replaces.
    struct FuseStats { size_t two_products = 0, multiply_adds = 0, pre_adds = 0; };
    bool fuse_products(uint32_t result, FuseStats& fs, bool pre_adds) {
        const size_t ni = words.size() / IW;
        std::vector<uint32_t> reads(n_virtual, 0), writes(n_virtual, 0), def(n_virtual, 0);
        for (size_t pc = 0; pc < ni; pc++) {
            const uint32_t* w = &words[pc * IW];
            for (int q = 2; q < 6; q++)
                if ((w[q] >> 28) == K_SLOT) reads[w[q] & 0x0fffffffu]++;
            if (w[1] != DST_ACC_V) {
                writes[w[1]]++;
                def[w[1]] = (uint32_t)pc;
            }
        }
        if ((result >> 28) == K_SLOT) reads[result & 0x0fffffffu]++;
        std::vector<uint8_t> dead(ni, 0);
        auto lone_product = [&](uint32_t opnd, size_t pc, size_t& src) {
            if ((opnd >> 28) != K_SLOT) return false;
            const uint32_t v = opnd & 0x0fffffffu;
            if (reads[v] != 1 || writes[v] != 1) return false;
            src = def[v];
            if (src >= pc || dead[src]) return false;
            const uint32_t* p = &words[src * IW];
            if ((p[0] & F_OPS) != F_MUL) return false;
            for (int q = 2; q < 4; q++) {
                const uint32_t kind = p[q] >> 28;
                if (kind == K_ACC) return false;
                if (kind == K_SLOT && writes[p[q] & 0x0fffffffu] != 1) return false;
            }
            return true;
        };
        for (size_t pc = 0; pc < ni; pc++) {
            uint32_t* w = &words[pc * IW];
            size_t src;
            if (!(w[0] & (F_ADD | F_SUB)) || (w[0] & (F_RSUB | F_MUL2))) continue;
            if (!(w[0] & F_MUL)) {
                if (lone_product(w[2], pc, src)) {             // p q +- c
                    w[0] = (w[0] & (F_ADD | F_SUB)) | F_MUL;
                    w[2] = words[src * IW + 2];
                    w[3] = words[src * IW + 3];
                } else if (lone_product(w[4], pc, src)) {      // x +- p q
                    const uint32_t x = w[2];
                    w[0] = F_MUL | ((w[0] & F_ADD) ? F_ADD : F_RSUB);
                    w[2] = words[src * IW + 2];
                    w[3] = words[src * IW + 3];
                    w[4] = x;
                } else {
                    continue;
                }
                dead[src] = 1;
                fs.multiply_adds++;
                if (w[0] & F_RSUB) continue;
            }
            if (lone_product(w[4], pc, src)) {                 // a b +- p q
                // the interpreter stages three memory operands: a must be a value slot or the running value
                auto in_memory = [](uint32_t o) { const uint32_t k = o >> 28; return k == K_COL || k == K_CONST || k == K_SCALAR; };
                uint32_t p = words[src * IW + 2], q = words[src * IW + 3];
                if (in_memory(w[2]) && !in_memory(w[3])) std::swap(w[2], w[3]);
                if (in_memory(w[2]) && (w[0] & F_ADD) && !(in_memory(p) && in_memory(q))) {   // p q + a b
                    if (in_memory(p)) std::swap(p, q);
                    std::swap(w[2], p);
                    std::swap(w[3], q);
                }
                if (in_memory(w[2])) continue;
                w[0] |= F_MUL2;
                w[4] = p;
                w[5] = q;
                dead[src] = 1;
                fs.two_products++;
            }
        }
        // A sum or difference x +- y of two operands read from memory (columns, constants, scalars: normalised digits), made
        // for a single consumer that is still a plain multiplication after the pass above, moves into that consumer:
        // a (x +- y) is one instruction, and the sum never travels through a value slot.
        auto in_mem = [](uint32_t o) { const uint32_t k = o >> 28; return k == K_COL || k == K_CONST || k == K_SCALAR; };
        auto lone_sum = [&](uint32_t opnd, size_t pc, size_t& src) {
            if ((opnd >> 28) != K_SLOT) return false;
            const uint32_t v = opnd & 0x0fffffffu;
            if (reads[v] != 1 || writes[v] != 1) return false;
            src = def[v];
            if (src >= pc || dead[src]) return false;
            const uint32_t* p = &words[src * IW];
            const uint32_t op = p[0] & F_OPS;
            if (op != F_ADD && op != F_SUB && op != F_RSUB) return false;
            return in_mem(p[2]) && in_mem(p[4]);
        };
        for (size_t pc = 0; pre_adds && pc < ni; pc++) {
            uint32_t* w = &words[pc * IW];
            if (dead[pc] || (w[0] & F_OPS) != F_MUL) continue;
            size_t src;
            if (!lone_sum(w[3], pc, src)) {
                if (!lone_sum(w[2], pc, src)) continue;
                std::swap(w[2], w[3]);
            }
            const uint32_t* p = &words[src * IW];
            const uint32_t op = p[0] & F_OPS;
            w[3] = op == F_RSUB ? p[4] : p[2];
            w[4] = op == F_RSUB ? p[2] : p[4];
            w[0] = F_MUL | F_PRE | (op == F_ADD ? 0u : F_PRESUB);
            dead[src] = 1;
            fs.pre_adds++;
        }
        // drop the moved instructions, then redo the magnitude bookkeeping over what is left
        size_t keep = 0;
        for (size_t pc = 0; pc < ni; pc++) {
            if (dead[pc]) continue;
            if (keep != pc) std::copy(&words[pc * IW], &words[pc * IW] + IW, &words[keep * IW]);
            keep++;
        }
        words.resize(keep * IW);
        std::fill(vbound.begin(), vbound.end(), 0.0);
        std::fill(vdig.begin(), vdig.end(), Digits());
        acc_bound = 0.0;
        acc_dig = Digits();
        for (size_t pc = 0; pc < keep; pc++) {
            uint32_t* w = &words[pc * IW];
            double v;
            w[0] = track(w[0], w[1], w[2], w[3], w[4], w[5], v);
            if (v > EV_LIMIT) {
                // allowed only where the emitter already put its multiplication by one right behind
                const uint32_t* nx = w + IW;
                const uint32_t self = w[1] == DST_ACC_V ? W_ACC : slot(w[1]);
                if (pc + 1 >= keep || (nx[0] & F_OPS) != F_MUL || nx[1] != w[1] || nx[2] != self || nx[3] != scalar(SC_ONE))
                    return false;
            }
        }
        return true;
    }
    // value-returning helpers (destination = fresh virtual value)
    uint32_t mul(uint32_t a, uint32_t b) { const uint32_t d = fresh(); emit(F_MUL, d, a, b); return slot(d); }
    uint32_t add(uint32_t a, uint32_t c) { const uint32_t d = fresh(); emit(F_ADD, d, a, W_ZERO, c); return slot(d); }
    uint32_t sub(uint32_t a, uint32_t c) { const uint32_t d = fresh(); emit(F_SUB, d, a, W_ZERO, c); return slot(d); }
    uint32_t mul_add(uint32_t a, uint32_t b, uint32_t c) { const uint32_t d = fresh(); emit(F_MUL | F_ADD, d, a, b, c); return slot(d); }
    uint32_t mul_sub(uint32_t a, uint32_t b, uint32_t c) { const uint32_t d = fresh(); emit(F_MUL | F_SUB, d, a, b, c); return slot(d); }

    // ---- the terms of h --------------------------------------------------------------------------------------
    // Upstream folds the N terms of the quotient numerator (gate polynomials, permutation and lookup identities) one
    // after the other, values[idx] = values[idx] * y + term_i, so that h = sum_i y^(N-1-i) term_i.  Nearly every term is
    // itself a product F * e with a factor shared by many terms: a selector (times a condition) in front of the
    // constraints of a gate, l_0 / l_last / l_active in front of the argument identities.  With the powers of y as
    // scalars of the call,
    //     h = sum over factors F of  F * (sum of y^(N-1-i) e_i over the terms i with that factor)  +  the other terms,
    // which is one product per term plus one per factor where the chain needs two per term; the products of a group go
    // out in pairs that share a reduction (a b + c d).  Terms without a shared factor keep the chain among themselves,
    // H = H * y^gap + term, which the peephole pass fuses with the term's own last product as before; H joins the sum
    // with the weight of its last term.  The field arithmetic is exact, so the value is the same.
    // The running-value register holds H while the gates are lowered, afterwards (or when no gate term is outside a
    // group) the partial sum of the one group that is open: a value in the register costs no slot.
    bool grouping = false;
    // which terms are issued (zkhip_evalh_desc::term_select): a term that is not kept still takes its number, so that every
    // kept term keeps its power of y and the low and the high program add up to the whole
    uint32_t term_select = 0, term_degree_max = 0;
    bool keep(uint32_t degree) const { return term_select == 0 || (term_select == 1) == (degree <= term_degree_max); }
    std::vector<std::pair<const char*, size_t>> marks;   // section boundaries (instruction index), for the debug print
    void mark(const char* name) { marks.push_back({name, words.size() / IW}); }
    uint32_t n_terms = 0, sc_y0 = 0;
    static constexpr uint32_t GAP_TAG = 1u << 24;      // scalar operands y^gap until finish_terms knows where they live
    std::vector<uint32_t> gaps;                        // exponents of the y^gap scalars, in order of first use
    struct PairSum {               // a sum of products under construction: pairs go out as they complete
        uint32_t sum = W_ZERO, px = 0, py = 0;
        bool has_sum = false, has_pend = false, in_acc = false;
    };
    std::map<uint32_t, PairSum> open_groups;   // by factor operand
    PairSum total;
    bool chain_reserved = false;   // the gates graph is being lowered and has terms outside groups: H is still growing
    bool chain_started = false, acc_taken = false, has_stash = false;
    uint32_t chain_last = 0, stash = 0;
    static bool in_memory(uint32_t o) { const uint32_t k = o >> 28; return k == K_COL || k == K_CONST || k == K_SCALAR; }
    uint32_t to_slot(uint32_t w) {
        const uint32_t v = fresh();
        emit(0, v, w);
        return slot(v);
    }
    // t = x y + px py as one instruction (a must be a value slot or the register: three memory operands are staged)
    void emit_pair(uint32_t dst, uint32_t a, uint32_t b2, uint32_t c, uint32_t d) {
        if (in_memory(a)) std::swap(a, b2);
        if (in_memory(a)) {
            std::swap(a, c);
            std::swap(b2, d);
            if (in_memory(a)) std::swap(a, b2);
        }
        if (in_memory(a)) a = to_slot(a);
        emit(F_MUL | F_ADD | F_MUL2, dst, a, b2, c, d);
    }
    void pair_push(PairSum& A, uint32_t x, uint32_t y) {
        if (!A.has_pend) {
            A.px = x;
            A.py = y;
            A.has_pend = true;
            return;
        }
        A.has_pend = false;
        if (A.in_acc && !A.has_sum) {
            emit_pair(DST_ACC_V, A.px, A.py, x, y);
        } else {
            const uint32_t t = fresh();
            emit_pair(t, A.px, A.py, x, y);
            if (A.in_acc) emit(F_ADD, DST_ACC_V, W_ACC, W_ZERO, slot(t));
            else A.sum = A.has_sum ? add(A.sum, slot(t)) : slot(t);
        }
        if (A.in_acc) A.sum = W_ACC;
        A.has_sum = true;
    }
    uint32_t pair_close(PairSum& A) {
        if (A.has_pend) {
            A.has_pend = false;
            if (A.in_acc) {
                if (A.has_sum) emit(F_MUL | F_ADD, DST_ACC_V, A.px, A.py, W_ACC);
                else emit(F_MUL, DST_ACC_V, A.px, A.py);
                A.sum = W_ACC;
            } else {
                A.sum = A.has_sum ? mul_add(A.px, A.py, A.sum) : mul(A.px, A.py);
            }
            A.has_sum = true;
        }
        return A.has_sum ? A.sum : W_ZERO;
    }
    // total += x * y with x in the register, which is about to be reused: nothing may stay pending on it
    void total_add_acc_times(uint32_t y) {
        if (total.has_pend) {
            pair_push(total, W_ACC, y);
        } else {
            total.sum = total.has_sum ? mul_add(W_ACC, y, total.sum) : mul(W_ACC, y);
            total.has_sum = true;
        }
    }
    // The terms are numbered in upstream's order; they may be issued in any order (terms outside groups: ascending).
    uint32_t reserve_terms(uint32_t k) {
        const uint32_t first = n_terms;
        n_terms += k;
        return first;
    }
    uint32_t gap_scalar(uint32_t g) {
        if (g == 1) return scalar(SC_Y);
        for (size_t q = 0; q < gaps.size(); q++)
            if (gaps[q] == g) return scalar(GAP_TAG + (uint32_t)q);
        gaps.push_back(g);
        return scalar(GAP_TAG + (uint32_t)gaps.size() - 1);
    }
    // term i = term (chain form: the next term of the chain, i is not looked at)
    void fold_at(uint32_t i, uint32_t term) {
        if (!grouping) {
            emit(F_MUL | F_ADD, DST_ACC_V, W_ACC, scalar(SC_Y), term);   // values[idx] = values[idx] * y + term
            return;
        }
        if (acc_taken) {                      // the register serves a group: this term joins the sum on its own
            pair_push(total, term, scalar(sc_y0 + i));
            return;
        }
        if (!chain_started) emit(0, DST_ACC_V, term);
        else emit(F_MUL | F_ADD, DST_ACC_V, W_ACC, gap_scalar(i - chain_last), term);
        chain_started = true;
        chain_last = i;
    }
    // H joins the sum; the register is free again
    void end_chain() {
        if (!chain_started) return;
        total_add_acc_times(scalar(sc_y0 + chain_last));
        chain_started = false;
    }
    // term i = e * factor
    void fold_times_at(uint32_t i, uint32_t e, uint32_t factor) {
        if (!grouping) {
            fold_at(i, mul(e, factor));
            return;
        }
        auto it = open_groups.find(factor);
        if (it == open_groups.end()) {
            it = open_groups.emplace(factor, PairSum()).first;
            if (!acc_taken) {
                // the register goes to this group's partial sum.  While the gates are lowered H steps aside into a slot
                // (it rests while the members of the group are issued back to back); afterwards H is complete
                if (chain_started && chain_reserved) {
                    stash = to_slot(W_ACC);
                    has_stash = true;
                } else {
                    end_chain();
                }
                it->second.in_acc = true;
                acc_taken = true;
            }
        }
        pair_push(it->second, e, scalar(sc_y0 + i));
    }
    void fold(uint32_t term) { fold_at(reserve_terms(1), term); }
    // no further term carries this factor
    void close_group(uint32_t factor) {
        auto it = open_groups.find(factor);
        if (it == open_groups.end()) return;
        const uint32_t s = pair_close(it->second);
        const bool in_acc = it->second.in_acc;
        open_groups.erase(it);
        if (in_acc) {
            acc_taken = false;
            if (s != W_ZERO) total_add_acc_times(factor);
            if (has_stash) {
                emit(0, DST_ACC_V, stash);
                has_stash = false;
            }
        } else if (s != W_ZERO) {
            pair_push(total, s, factor);
        }
    }
    // the operand that holds h once every term is in
    uint32_t finish_terms() {
        if (!grouping) return W_ACC;
        while (!open_groups.empty()) close_group(open_groups.begin()->first);
        uint32_t result;
        if (chain_started && !total.has_sum && !total.has_pend && chain_last + 1 == n_terms) {
            result = W_ACC;                   // nothing but the chain, its last term weighs y^0
        } else {
            end_chain();
            result = pair_close(total);
        }
        // the y^gap scalars follow the n_terms weights
        for (size_t q = 0; q + IW <= words.size(); q += IW)
            for (size_t k = 2; k < IW; k++) {
                uint32_t& w = words[q + k];
                if ((w >> 28) == K_SCALAR && (w & 0x0fffffffu) >= GAP_TAG) w = scalar(sc_y0 + n_terms + ((w & 0x0fffffffu) - GAP_TAG));
            }
        return result;
    }
};

// Lowers one GraphEvaluator into the builder.  `is_gates`: the root Horner accumulates into the running value.
struct Lowering {
    Builder& b;
    const zkhip_graph* g;
    const zkhip_evalh_desc* desc;
    bool is_gates;
    std::vector<Calc> calcs;
    std::vector<uint32_t> val;
    std::vector<uint8_t> done;
    uint32_t const_base = 0;
    uint32_t root = 0;
    std::string err;

    Lowering(Builder& b_, const zkhip_graph* g_, const zkhip_evalh_desc* d_, bool gates) : b(b_), g(g_), desc(d_), is_gates(gates) {}

    bool operand(const VS& v, uint32_t& word) {
        switch (v.kind) {
            case 0:
                if (v.a >= g->num_constants) { err = "constant index out of range"; return false; }
                word = (K_CONST << 28) | (const_base + v.a);
                return true;
            case 1:
                if (v.a >= calcs.size() || !done[v.a]) { err = "intermediate used before definition"; return false; }
                word = val[v.a];
                return true;
            case 2: case 3: case 4: {
                uint32_t c;
                if (v.kind == 2) { if (v.a >= desc->num_fixed) { err = "fixed column out of range"; return false; } c = v.a; }
                else if (v.kind == 3) { if (v.a >= desc->num_advice) { err = "advice column out of range"; return false; } c = desc->num_fixed + v.a; }
                else { if (v.a >= desc->num_instance) { err = "instance column out of range"; return false; } c = desc->num_fixed + desc->num_advice + v.a; }
                if (v.b >= g->num_rotations) { err = "rotation index out of range"; return false; }
                word = b.col(c, g->rotations[v.b]);
                return true;
            }
            case 5:
                if (v.a >= desc->num_challenges) { err = "challenge index out of range"; return false; }
                word = Builder::scalar(SC_CHALLENGE0 + v.a);
                return true;
            case 6: word = Builder::scalar(SC_BETA); return true;
            case 7: word = Builder::scalar(SC_GAMMA); return true;
            case 8: word = Builder::scalar(SC_THETA); return true;
            case 9: word = Builder::scalar(SC_Y); return true;
            case 10: word = is_gates ? W_ACC : W_ZERO; return true;   // previous_value: values[idx] for gates, ZERO for lookups
            default: err = "unknown ValueSource kind"; return false;
        }
    }

    // degree of every calculation as a polynomial in the columns (constants, challenges and the running value: 0)
    std::vector<uint32_t> deg;
    uint32_t vs_degree(const VS& v) const {
        if (v.kind == 1) return v.a < deg.size() ? deg[v.a] : 0;
        return (v.kind >= 2 && v.kind <= 4) ? 1 : 0;
    }
    void compute_degrees() {
        deg.assign(calcs.size(), 0);
        for (size_t t = 0; t < calcs.size(); t++) {
            const Calc& c = calcs[t];
            switch (c.op) {
                case 0: case 1: deg[t] = std::max(vs_degree(c.src[0]), vs_degree(c.src[1])); break;
                case 2: deg[t] = vs_degree(c.src[0]) + vs_degree(c.src[1]); break;
                case 3: deg[t] = 2 * vs_degree(c.src[0]); break;
                case 6: {
                    uint32_t d = vs_degree(c.start);
                    const uint32_t f = vs_degree(c.factor);
                    for (const VS& p : c.parts) d = std::max(d + f, vs_degree(p));
                    deg[t] = d;
                    break;
                }
                default: deg[t] = vs_degree(c.src[0]); break;
            }
        }
    }
    bool whole_skipped = false;   // a gates graph that is one term of h, and that term is not kept

    // iterative post-order: make calculation `t_root` available
    bool schedule(uint32_t t_root) {
        struct Frame { uint32_t t; uint32_t stage; };
        std::vector<Frame> st;
        st.push_back({t_root, 0});
        while (!st.empty()) {
            const uint32_t t = st.back().t;
            if (done[t]) { st.pop_back(); continue; }
            Calc& c = calcs[t];
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
                uint32_t x, y = W_ZERO;
                if (!operand(c.src[0], x)) return false;
                if (ns == 2 && !operand(c.src[1], y)) return false;
                switch (c.op) {
                    case 0: val[t] = b.add(x, y); break;
                    case 1: val[t] = b.sub(x, y); break;
                    case 2: val[t] = b.mul(x, y); break;
                    case 3: val[t] = b.mul(x, x); break;
                    case 4: val[t] = b.add(x, x); break;
                    case 5: val[t] = b.sub(W_ZERO, x); break;
                    default: val[t] = x; break;   // Store: alias the source
                }
                done[t] = 1;
                st.pop_back();
            } else if (terms_root && t == root) {
                // the root of the gates graph, previous_value * y^k + sum parts_j y^(k-1-j): every part is a term of h
                if (st.back().stage == 0) {
                    plan_root_terms(c);
                    st.back().stage = 1;
                }
                bool pushed = false;
                while (st.back().stage - 1 < root_order.size()) {
                    const uint32_t j = root_order[st.back().stage - 1];
                    const RootPart& rp = root_parts[j];
                    if (!b.keep(vs_degree(c.parts[j]))) {      // the term keeps its number and nothing of it is issued
                        st.back().stage++;                        // (groups are formed among the kept parts only)
                        continue;
                    }
                    if (rp.group < 0) {
                        if (dep(c.parts[j])) { pushed = true; break; }
                        uint32_t pw;
                        if (!operand(c.parts[j], pw)) return false;
                        b.fold_at(root_term0 + j, pw);
                    } else {
                        if (dep(rp.factor) || dep(rp.other)) { pushed = true; break; }
                        uint32_t fw, ew;
                        if (!operand(rp.factor, fw) || !operand(rp.other, ew)) return false;
                        b.fold_times_at(root_term0 + j, ew, fw);
                        if (--group_left[rp.group] == 0) b.close_group(fw);
                    }
                    st.back().stage++;
                }
                if (!err.empty()) return false;
                if (pushed) continue;
                val[t] = W_ZERO;        // the value lives in the builder's terms
                done[t] = 1;
                st.pop_back();
            } else {
                // Horner: stage 0 = start / factor dependencies, then one part at a time
                const bool to_acc = is_gates && t == root && !b.grouping;
                if (st.back().stage == 0) {
                    if (dep(c.start) || dep(c.factor)) { if (!err.empty()) return false; continue; }
                    uint32_t s;
                    if (!operand(c.start, s)) return false;
                    c.acc_dst = to_acc ? DST_ACC_V : b.fresh();
                    if (!(to_acc && s == W_ACC)) b.emit(0, c.acc_dst, s);   // the running value already holds previous_value
                    st.back().stage = 1;
                }
                const uint32_t accw = (c.acc_dst == DST_ACC_V) ? W_ACC : Builder::slot(c.acc_dst);
                bool pushed = false;
                while (st.back().stage - 1 < c.parts.size()) {
                    const VS& p = c.parts[st.back().stage - 1];
                    if (dep(p)) { pushed = true; break; }
                    uint32_t pw, fw;
                    if (!operand(p, pw) || !operand(c.factor, fw)) return false;
                    b.emit(F_MUL | F_ADD, c.acc_dst, accw, fw, pw);
                    st.back().stage++;
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

    // ---- terms of the root Horner ------------------------------------------------------------------------
    // A part that is a product nobody else reads, F * e, joins the group of its factor F when another part shares that
    // factor; the factor with the most parts is served first (ties: the lower key), so that `q * cond` in front of the
    // constraints of a gate wins over a column that happens to occur twice.
    struct RootPart {
        int32_t group = -1;
        VS factor{0, 0, 0}, other{0, 0, 0};
    };
    bool terms_root = false;
    std::vector<RootPart> root_parts;
    std::vector<uint32_t> group_left;
    std::vector<uint32_t> root_order;   // the parts in the order they are issued: the members of a group back to back, so
                                        // that one partial sum is live at a time
    uint32_t root_term0 = 0;
    static uint64_t vs_key(const VS& v) { return ((uint64_t)v.kind << 56) ^ ((uint64_t)v.a << 24) ^ (uint64_t)v.b; }
    void plan_root_terms(const Calc& c) {
        const size_t np = c.parts.size();
        root_parts.assign(np, RootPart());
        // how often every calculation is read
        std::vector<uint32_t> uses(calcs.size(), 0);
        auto use = [&](const VS& v) { if (v.kind == 1 && v.a < calcs.size()) uses[v.a]++; };
        for (const Calc& k : calcs) {
            if (k.op == 6) {
                use(k.start);
                use(k.factor);
                for (const VS& v : k.parts) use(v);
            } else {
                use(k.src[0]);
                if (k.op <= 2) use(k.src[1]);
            }
        }
        struct Cand { size_t part; VS f[2]; };
        std::vector<Cand> cand;
        for (size_t j = 0; j < np; j++) {
            const VS& v = c.parts[j];
            if (v.kind != 1 || v.a >= calcs.size() || uses[v.a] != 1 || !b.keep(vs_degree(v))) continue;
            const Calc& m = calcs[v.a];
            if (m.op != 2 || vs_key(m.src[0]) == vs_key(m.src[1])) continue;
            cand.push_back({j, {m.src[0], m.src[1]}});
        }
        std::vector<uint8_t> taken(cand.size(), 0);
        for (;;) {
            std::map<uint64_t, uint32_t> count;
            for (size_t q = 0; q < cand.size(); q++)
                if (!taken[q]) { count[vs_key(cand[q].f[0])]++; count[vs_key(cand[q].f[1])]++; }
            uint64_t best = 0;
            uint32_t best_n = 1;
            for (const auto& kv : count)
                if (kv.second > best_n) { best = kv.first; best_n = kv.second; }
            if (best_n < 2) break;
            const int32_t gid = (int32_t)group_left.size();
            group_left.push_back(0);
            for (size_t q = 0; q < cand.size(); q++) {
                if (taken[q]) continue;
                const int side = vs_key(cand[q].f[0]) == best ? 0 : (vs_key(cand[q].f[1]) == best ? 1 : -1);
                if (side < 0) continue;
                RootPart& rp = root_parts[cand[q].part];
                rp.group = gid;
                rp.factor = cand[q].f[side];
                rp.other = cand[q].f[1 - side];
                group_left[gid]++;
                taken[q] = 1;
            }
        }
        root_term0 = b.reserve_terms((uint32_t)np);
        for (size_t j = 0; j < np; j++)
            if (root_parts[j].group < 0 && b.keep(vs_degree(c.parts[j]))) b.chain_reserved = true;   // the register carries the chain of those parts
        std::vector<std::vector<uint32_t>> members(group_left.size());
        for (size_t j = 0; j < np; j++)
            if (root_parts[j].group >= 0) members[root_parts[j].group].push_back((uint32_t)j);
        std::vector<uint8_t> issued(group_left.size(), 0);
        root_order.clear();
        for (size_t j = 0; j < np; j++) {
            const int32_t gi = root_parts[j].group;
            if (gi < 0) {
                root_order.push_back((uint32_t)j);
            } else if (!issued[gi]) {
                issued[gi] = 1;
                root_order.insert(root_order.end(), members[gi].begin(), members[gi].end());
            }
        }
    }

    // returns the operand word of the graph's value (W_ACC for a gates graph rooted in a Horner)
    bool run(uint32_t& result) {
        if (!parse_graph(g, calcs, err)) return false;
        const_base = (uint32_t)(b.consts.size() / 4);
        if (g->num_constants) b.consts.insert(b.consts.end(), g->constants, g->constants + (size_t)g->num_constants * 4);
        if (calcs.empty()) { result = W_ZERO; return true; }
        val.assign(calcs.size(), 0);
        done.assign(calcs.size(), 0);
        root = (uint32_t)calcs.size() - 1;
        terms_root = is_gates && b.grouping && calcs[root].op == 6 && calcs[root].start.kind == 10 && calcs[root].factor.kind == 9;
        compute_degrees();
        if (is_gates && b.grouping && !terms_root && !b.keep(deg[root])) {
            whole_skipped = true;
            result = W_ZERO;
            return true;
        }
        if (!schedule(root)) return false;
        result = val[root];
        return true;
    }
};

// Liveness-based slot assignment.  lds_budget: LDS slots one lane may use; the program gets min(peak, budget).
static void assign_slots(std::vector<uint32_t>& words, uint32_t& result, uint32_t n_virtual, uint32_t lds_budget,
                         uint32_t& lds_slots_out, uint32_t& n_slots_out) {
    const size_t ni = words.size() / IW;
    std::vector<uint32_t> first(n_virtual, UINT32_MAX), last(n_virtual, 0), uses(n_virtual, 0);
    auto touch = [&](uint32_t v, uint32_t pc) {
        if (first[v] == UINT32_MAX) first[v] = pc;
        last[v] = std::max(last[v], pc);
        uses[v]++;
    };
    for (size_t pc = 0; pc < ni; pc++) {
        const uint32_t dst = words[IW * pc + 1];
        for (size_t k = 2; k < IW; k++)
            if ((words[IW * pc + k] >> 28) == K_SLOT) touch(words[IW * pc + k] & 0x0fffffffu, (uint32_t)pc);
        if (dst != DST_ACC_V) touch(dst, (uint32_t)pc);
    }
    if ((result >> 28) == K_SLOT) touch(result & 0x0fffffffu, (uint32_t)ni);
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
        // smallest prefix of `order` (largest access gaps first) whose demotion fits the rest on chip
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
    std::vector<uint32_t> phys(n_virtual, 0);
    std::vector<std::vector<uint32_t>> expire(ni + 2), begin(ni + 2);
    for (uint32_t v = 0; v < n_virtual; v++)
        if (first[v] != UINT32_MAX) {
            expire[last[v] + 1].push_back(v);
            begin[first[v]].push_back(v);
        }
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
        const uint32_t dst = words[IW * pc + 1];
        if (dst != DST_ACC_V) words[IW * pc + 1] = phys[dst];
        for (size_t k = 2; k < IW; k++) remap(words[IW * pc + k]);
    }
    remap(result);
    n_slots_out = lds_slots + next_sp;
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
        if (count) ZK_HIP(ctx, hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        *out = (T*)d;
        return ZKHIP_OK;
    }
};

static bool graph_arrays_ok(const zkhip_graph* g) {
    return (g->constants || !g->num_constants) && (g->rotations || !g->num_rotations) && (g->calc_words || !g->num_calc_words);
}

// The fused program of one constraint-system shape, slots assigned, operands still symbolic (column numbers, rotation
// indices).  It depends on the graphs and on the counts of the descriptor only -- not on challenges, column pointers or
// the coset list -- so it is built once per shape and kept in the context (upstream builds `Evaluator` once per proving
// key); a call links it against its column pointers, which is one pass over the operands.
struct EvalhProgram {
    std::vector<uint32_t> words;     // IW words per instruction
    std::vector<uint64_t> consts;    // merged constant table, already in the 2^261 Montgomery domain
    std::vector<int32_t> rots;
    uint32_t result = W_ACC, lds_slots = 1, n_slots = 1, n_cols = 0;
    uint32_t n_terms = 0, sc_y0 = 0;     // scalars sc_y0 + i hold y^(n_terms - 1 - i): the weight of term i of h
    std::vector<uint32_t> gaps;          // followed by y^gaps[q]
    uint64_t stats[4] = {0, 0, 0, 0};
    std::vector<uint8_t> col_used;       // per column of the unified table: does any instruction read it
};

// Position of every column group in the unified column table:
// fixed | advice | instance | l0 l_last l_active | X | perm z | perm sigma | per lookup (z, a', s')
struct ColLayout {
    uint32_t c_l0 = 0, c_x = 0, c_z0 = 0, c_sigma0 = 0, c_lk0 = 0, total = 0;
    ColLayout(const zkhip_evalh_desc* d, bool lagrange) {
        uint32_t c = d->num_fixed + d->num_advice + d->num_instance;
        c_l0 = c;
        if (!lagrange) c += 3;
        if (d->num_perm_sets) {
            c_x = c++;
            c_z0 = c;
            c += d->num_perm_sets;
            c_sigma0 = c;
            c += d->num_perm_columns;
        }
        c_lk0 = c;
        total = c + 3 * d->num_lookups;
    }
};

// two independent 64-bit mixes over everything the program depends on
struct ShapeHash {
    uint64_t a = 0x9E3779B97F4A7C15ull, b = 0xC2B2AE3D27D4EB4Full;
    void word(uint64_t w) {
        a = (a ^ w) * 0x100000001B3ull;
        a ^= a >> 29;
        b = (b + w + 0x632BE59BD9B4E019ull) * 0xD6E8FEB86659FD93ull;
        b ^= b >> 32;
    }
    void bytes(const void* p, size_t n) {
        const unsigned char* q = (const unsigned char*)p;
        word(n);
        size_t i = 0;
        for (; i + 8 <= n; i += 8) {
            uint64_t w;
            memcpy(&w, q + i, 8);
            word(w);
        }
        uint64_t w = 0;
        if (i < n) {
            memcpy(&w, q + i, n - i);
            word(w);
        }
    }
    void graph(const zkhip_graph* g) {
        word(g->num_calculations);
        bytes(g->constants, (size_t)g->num_constants * 32);
        bytes(g->rotations, (size_t)g->num_rotations * 4);
        bytes(g->calc_words, (size_t)g->num_calc_words * 4);
    }
};

static int build_program(zkhip_ctx* ctx, const zkhip_evalh_desc* desc, bool lagrange, uint32_t lds_budget, bool do_fuse,
                         bool pre_adds, EvalhProgram& out) {
    const ColLayout L(desc, lagrange);
    const uint32_t c_l0 = L.c_l0, c_last = L.c_l0 + 1, c_active = L.c_l0 + 2;
    Builder b;
    const uint32_t ONE = Builder::scalar(SC_ONE), BETA = Builder::scalar(SC_BETA), GAMMA = Builder::scalar(SC_GAMMA);
    // scalars after the challenges: beta * delta^c per permutation column, then the powers of y (filled in per call)
    const uint32_t sc_perm0 = SC_CHALLENGE0 + desc->num_challenges;
    // ZKHIP_EVALH_CHAIN: the terms as upstream's chain h = h * y + term (measurements and differential tests)
    static const bool chain = getenv("ZKHIP_EVALH_CHAIN") != nullptr;
    b.grouping = !chain;
    b.term_select = lagrange ? 0 : desc->term_select;
    b.term_degree_max = desc->term_degree_max;
    ZK_CHECK(ctx, !b.term_select || b.grouping, "evaluate_h: a term selection needs the grouped form (ZKHIP_EVALH_CHAIN is set)");
    b.sc_y0 = sc_perm0 + (desc->num_perm_sets ? desc->num_perm_columns : 0);
    // ---- 1. custom gates -----------------------------------------------------------------------------
    if (desc->custom_gates) {
        const zkhip_graph* g = desc->custom_gates;
        Lowering lw(b, g, desc, true);
        uint32_t res;
        if (!lw.run(res)) return set_error(ctx, ZKHIP_ERR_INVALID, "evaluate_h: %s", lw.err.c_str());
        if (b.grouping) {
            if (!lw.terms_root) {                      // the value of the graph is the first term
                const uint32_t i0 = b.reserve_terms(1);
                if (!lw.whole_skipped) b.fold_at(i0, res);
            }
            b.chain_reserved = false;                  // the groups that follow may use the register
        } else if (res != W_ACC) {
            b.emit(0, DST_ACC_V, res);                 // a gates graph whose value is not the running Horner
        }
    }

    // ---- 2. permutation argument ---------------------------------------------------------------------
    b.mark("gates");
    // ---- 2. + 3. permutation and lookup arguments ---------------------------------------------------------
    // Every identity is (expression) * l_0, l_last or l_active.  The terms get their numbers in upstream's order
    // (permutation argument, then lookup after lookup); they are issued factor by factor, so that one partial sum is
    // live at a time (a lookup's a' - s' is then formed twice, a subtraction).
    struct Term {
        uint32_t idx;
        std::function<int(uint32_t&)> make;    // emits the expression, returns its operand
    };
    std::vector<Term> by_factor[3];            // 0: l_0, 1: l_last, 2: l_active
    const uint32_t l_cols[3] = {c_l0, c_last, c_active};
    // degree: of the whole term as a polynomial in the columns, its l_0 / l_last / l_active factor included
    auto term = [&](int f, uint32_t degree, std::function<int(uint32_t&)> make) {
        const uint32_t idx = b.reserve_terms(1);
        if (b.keep(degree)) by_factor[f].push_back({idx, std::move(make)});
    };
    std::vector<uint32_t> pcol(desc->num_perm_sets ? desc->num_perm_columns : 0);
    if (desc->num_perm_sets) {
        const uint32_t c_x = L.c_x, c_z0 = L.c_z0, c_sigma0 = L.c_sigma0;
        for (uint32_t c = 0; c < desc->num_perm_columns; c++) {
            const uint32_t kind = desc->perm_column_kind[c], ci = desc->perm_column_index[c];
            if (kind == 2) { ZK_CHECK(ctx, ci < desc->num_fixed, "evaluate_h: permutation fixed column out of range"); pcol[c] = ci; }
            else if (kind == 3) { ZK_CHECK(ctx, ci < desc->num_advice, "evaluate_h: permutation advice column out of range"); pcol[c] = desc->num_fixed + ci; }
            else if (kind == 4) { ZK_CHECK(ctx, ci < desc->num_instance, "evaluate_h: permutation instance column out of range"); pcol[c] = desc->num_fixed + desc->num_advice + ci; }
            else return set_error(ctx, ZKHIP_ERR_INVALID, "evaluate_h: permutation column kind %u", kind);
        }
        const int32_t last_rot = -((int32_t)desc->blinding_factors + 1);
        const uint32_t nsets = desc->num_perm_sets;
        // l_0(X) * (1 - z_0(X))
        term(0, 2, [&b, ONE, c_z0](uint32_t& e) { e = b.sub(ONE, b.col(c_z0, 0)); return (int)ZKHIP_OK; });
        // l_last(X) * (z_l(X)^2 - z_l(X))
        term(1, 3, [&b, c_z0, nsets](uint32_t& e) {
            const uint32_t zl = b.col(c_z0 + nsets - 1, 0);
            e = b.mul_sub(zl, zl, zl);
            return (int)ZKHIP_OK;
        });
        // l_0(X) * (z_i(X) - z_{i-1}(omega^last X))
        for (uint32_t s = 1; s < nsets; s++)
            term(0, 2, [&b, c_z0, s, last_rot](uint32_t& e) {
                e = b.sub(b.col(c_z0 + s, 0), b.col(c_z0 + s - 1, last_rot));
                return (int)ZKHIP_OK;
            });
        // l_active(X) * (z_i(omega X) prod (p + beta s_j + gamma) - z_i(X) prod (p + delta^j beta X + gamma));
        // current_delta of column c on row idx = beta * zeta * extended_omega^idx * delta^c = (beta * delta^c) * X[idx]
        for (uint32_t s = 0; s < nsets; s++)
            term(2, 2 + std::min(desc->perm_chunk_len, desc->num_perm_columns - s * desc->perm_chunk_len),
                 [&b, &pcol, desc, s, c_z0, c_x, c_sigma0, sc_perm0, BETA, GAMMA](uint32_t& e) {
                const uint32_t c0 = s * desc->perm_chunk_len, c1 = std::min(c0 + desc->perm_chunk_len, desc->num_perm_columns);
                uint32_t left = b.col(c_z0 + s, 1), right = b.col(c_z0 + s, 0);
                for (uint32_t c = c0; c < c1; c++) {
                    const uint32_t val = b.col(pcol[c], 0);
                    left = b.mul(left, b.add(b.mul_add(BETA, b.col(c_sigma0 + c, 0), val), GAMMA));
                    right = b.mul(right, b.add(b.mul_add(b.col(c_x, 0), Builder::scalar(sc_perm0 + c), val), GAMMA));
                }
                e = b.sub(left, right);
                return (int)ZKHIP_OK;
            });
    }
    for (uint32_t i = 0; i < desc->num_lookups; i++) {
        const zkhip_graph* g = &desc->lookups[i].graph;
        const uint32_t c_z = L.c_lk0 + 3 * i, c_a = c_z + 1, c_s = c_z + 2;
        // l_0(X) * (1 - z(X))
        term(0, 2, [&b, ONE, c_z](uint32_t& e) { e = b.sub(ONE, b.col(c_z, 0)); return (int)ZKHIP_OK; });
        // l_last(X) * (z(X)^2 - z(X))
        term(1, 3, [&b, c_z](uint32_t& e) {
            const uint32_t Z = b.col(c_z, 0);
            e = b.mul_sub(Z, Z, Z);
            return (int)ZKHIP_OK;
        });
        // l_active(X) * (z(omega X) (a'(X) + beta) (s'(X) + gamma) - z(X) * table_value)
        uint32_t table_degree = 0;     // of (a_compressed + beta) (s_compressed + gamma), the graph's value
        {
            Lowering probe(b, g, desc, false);
            if (!parse_graph(g, probe.calcs, probe.err)) return set_error(ctx, ZKHIP_ERR_INVALID, "evaluate_h: %s", probe.err.c_str());
            probe.compute_degrees();
            if (!probe.calcs.empty()) table_degree = probe.deg.back();
        }
        term(2, 1 + std::max<uint32_t>(3, 1 + table_degree), [&b, ctx, g, desc, c_z, c_a, c_s, BETA, GAMMA](uint32_t& e) {
            Lowering lw(b, g, desc, false);
            uint32_t table_value;
            if (!lw.run(table_value)) return set_error(ctx, ZKHIP_ERR_INVALID, "evaluate_h: %s", lw.err.c_str());
            const uint32_t left = b.mul(b.mul(b.col(c_z, 1), b.add(b.col(c_a, 0), BETA)), b.add(b.col(c_s, 0), GAMMA));
            e = b.sub(left, b.mul(b.col(c_z, 0), table_value));
            return (int)ZKHIP_OK;
        });
        // l_0(X) * (a'(X) - s'(X))
        term(0, 2, [&b, c_a, c_s](uint32_t& e) { e = b.sub(b.col(c_a, 0), b.col(c_s, 0)); return (int)ZKHIP_OK; });
        // l_active(X) * (a'(X) - s'(X)) * (a'(X) - a'(omega^-1 X))
        term(2, 3, [&b, c_a, c_s](uint32_t& e) {
            const uint32_t A = b.col(c_a, 0);
            e = b.mul(b.sub(A, b.col(c_s, 0)), b.sub(A, b.col(c_a, -1)));
            return (int)ZKHIP_OK;
        });
    }
    if (b.grouping) {
        for (int f = 0; f < 3; f++) {
            for (const Term& t : by_factor[f]) {
                uint32_t e;
                const int rc = t.make(e);
                if (rc) return rc;
                b.fold_times_at(t.idx, e, b.col(l_cols[f], 0));
            }
            if (!by_factor[f].empty()) b.close_group(b.col(l_cols[f], 0));
            b.mark(f == 0 ? "l_0 terms" : f == 1 ? "l_last terms" : "l_active terms");
        }
    } else {
        // the chain takes its terms in upstream's order
        std::vector<std::pair<uint32_t, int>> seq;
        std::vector<const Term*> where;
        for (int f = 0; f < 3; f++)
            for (const Term& t : by_factor[f]) {
                seq.push_back({t.idx, f});
                where.push_back(&t);
            }
        std::vector<size_t> ord(seq.size());
        for (size_t q = 0; q < ord.size(); q++) ord[q] = q;
        std::sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return seq[x].first < seq[y].first; });
        for (size_t q : ord) {
            uint32_t e;
            const int rc = where[q]->make(e);
            if (rc) return rc;
            b.fold_times_at(seq[q].first, e, b.col(l_cols[seq[q].second], 0));
        }
    }
    ZK_CHECK(ctx, L.total < (1u << 16) && b.rots.size() < (1u << 12), "evaluate_h: too many columns or rotations");
    ZK_CHECK(ctx, b.n_virtual < (1u << 28), "evaluate_h: program too large");

    // ---- peephole fusion, slots -----------------------------------------------------------------------
    uint32_t result = b.finish_terms(), lds_slots = 1, n_slots = 1;
    ZK_CHECK(ctx, b.n_virtual < (1u << 28), "evaluate_h: program too large");
    Builder::FuseStats fuse;
    if (do_fuse) ZK_CHECK(ctx, b.fuse_products(result, fuse, pre_adds), "evaluate_h: internal error, a fused value left its bound");
    assign_slots(b.words, result, b.n_virtual, lds_budget, lds_slots, n_slots);
    // one multiplication = one product of two operands (an instruction with a second product counts two)
    size_t nmul = 0, ncol = 0, nspill = 0;
    const size_t ni = b.words.size() / IW;
    for (size_t pc = 0; pc < ni; pc++) {
        const uint32_t* w = &b.words[pc * IW];
        ZK_CHECK(ctx, w[1] == DST_ACC_V || w[1] < DST_ACC, "evaluate_h: too many value slots");
        nmul += ((w[0] & F_MUL) != 0) + ((w[0] & F_MUL2) != 0);
        for (size_t q = 2; q < IW; q++) {
            if ((w[q] >> 28) == K_COL) ncol++;
            if ((w[q] >> 28) == K_SLOT && (w[q] & 0x0fffffffu) >= lds_slots) nspill++;
        }
        if (w[1] != DST_ACC_V && w[1] >= lds_slots) nspill++;
    }
    if (getenv("ZKHIP_DEBUG") && ni > 100) {
        // spill accesses per section (the marks are instruction indices before the peephole pass: approximate)
        size_t from = 0;
        for (const auto& m : b.marks) {
            size_t acc = 0;
            const size_t to = std::min(ni, m.second);
            for (size_t pc = from; pc < to; pc++) {
                const uint32_t* w = &b.words[pc * IW];
                for (size_t q = 2; q < IW; q++)
                    if ((w[q] >> 28) == K_SLOT && (w[q] & 0x0fffffffu) >= lds_slots) acc++;
                if (w[1] != DST_ACC_V && w[1] >= lds_slots) acc++;
            }
            fprintf(stderr, "[zkhip]   %s: instructions [%zu, %zu), %zu spill accesses\n", m.first, from, to, acc);
            from = to;
        }
    }
    if (getenv("ZKHIP_DEBUG") && ni > 100) {
        // instruction shapes: flags, destination kind, operand kinds
        std::map<std::string, size_t> shapes;
        auto kind_of = [&](uint32_t w) -> char {
            const uint32_t k = w >> 28, pl = w & 0x0fffffffu;
            switch (k) {
                case K_CONST: case K_SCALAR: return 'u';
                case K_COL: return 'r';
                case K_SLOT: return pl == 0 ? '0' : (pl < lds_slots ? 'l' : 's');
                case K_ACC: return 'A';
                default: return 'z';
            }
        };
        for (size_t pc = 0; pc < ni; pc++) {
            const uint32_t* w = &b.words[pc * IW];
            char buf[64];
            const char dk = w[1] == DST_ACC_V ? 'A' : (w[1] == 0 ? '0' : (w[1] < lds_slots ? 'l' : 's'));
            snprintf(buf, sizeof buf, "%s%s%s dst=%c a=%c b=%c c=%c d=%c", (w[0] & F_MUL2) ? "MUL2" : ((w[0] & F_PRE) ? "MULP" : ((w[0] & F_MUL) ? "MUL " : "    ")),
                     (w[0] & F_ADD) ? "+" : "", (w[0] & (F_SUB | F_RSUB)) ? "-" : "", dk, kind_of(w[2]), kind_of(w[3]), kind_of(w[4]),
                     kind_of(w[5]));
            shapes[buf]++;
        }
        std::vector<std::pair<size_t, std::string>> order;
        for (auto& kv : shapes) order.push_back({kv.second, kv.first});
        std::sort(order.rbegin(), order.rend());
        for (size_t i = 0; i < order.size() && i < 25; i++) fprintf(stderr, "[zkhip]   %6zu  %s\n", order[i].first, order[i].second.c_str());
    }
    if (getenv("ZKHIP_DEBUG"))
        fprintf(stderr, "[zkhip] evaluate_h program: %zu instructions (%zu products, %zu of them second products sharing a "
                        "reduction, %zu multiply-adds formed, %zu sums moved into a product; %zu column reads, %zu spill accesses), %u columns, "
                        "%zu rotations, %u values -> %u slots (%u in LDS)\n",
                ni, nmul, fuse.two_products, fuse.multiply_adds, fuse.pre_adds, ncol, nspill, L.total, b.rots.size(), b.n_virtual,
                n_slots, lds_slots);
    // constants move to the interpreter's 2^261 Montgomery domain: x R -> 32 x R
    {
        uint64_t k32[4];
        memcpy(k32, FrParams::TWO5, 32);
        for (size_t i = 0; i + 4 <= b.consts.size(); i += 4) h_mul(HFR, &b.consts[i], &b.consts[i], k32);
    }
    out.col_used.assign(L.total, 0);
    for (size_t pc = 0; pc < ni; pc++)
        for (size_t q = 2; q < IW; q++) {
            const uint32_t w = b.words[pc * IW + q];
            if ((w >> 28) == K_COL) out.col_used[(w & 0x0fffffffu) >> 12] = 1;
        }
    if ((result >> 28) == K_COL) out.col_used[(result & 0x0fffffffu) >> 12] = 1;
    out.words.swap(b.words);
    out.consts.swap(b.consts);
    out.rots.swap(b.rots);
    out.result = result;
    out.lds_slots = lds_slots;
    out.n_slots = n_slots;
    out.n_cols = L.total;
    out.n_terms = b.grouping ? b.n_terms : 0;
    out.sc_y0 = b.sc_y0;
    out.gaps = b.gaps;
    out.stats[0] = ni;
    out.stats[1] = nmul;
    out.stats[2] = ncol;
    out.stats[3] = n_slots;
    return ZKHIP_OK;
}

// grow-only device buffers of the context for the linked program, its scalars and constants and the spilled value slots:
// after the first call of a shape, evaluate_h over device-resident extended columns allocates nothing
static int grow(zkhip_ctx* ctx, void*& p, size_t& have, size_t need) {
    if (need <= have) return ZKHIP_OK;
    if (p) {
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ZK_HIP(ctx, hipFree(p));
        p = nullptr;
        have = 0;
    }
    need = (need + need / 4 + 0xffff) & ~(size_t)0xffff;
    ZK_HIP(ctx, hipMalloc(&p, need));
    have = need;
    return ZKHIP_OK;
}

// The program of the descriptor's shape: from the context's cache, or built now and kept.
static int get_program(zkhip_ctx* ctx, const zkhip_evalh_desc* desc, bool lagrange, std::shared_ptr<EvalhProgram>& prog) {
    const char* env_slots = getenv("ZKHIP_EVALH_LDS_SLOTS");
    // on-chip budget per lane: one register slot + four LDS slots of 36 B (nine digits): four 256-lane workgroups per CU
    const uint32_t lds_budget = env_slots && *env_slots ? (uint32_t)atoi(env_slots) : 5;
    const char* env_fuse = getenv("ZKHIP_EVALH_FUSE");   // "0" keeps every product an instruction of its own
    const bool do_fuse = !(env_fuse && *env_fuse == '0');
    const char* env_pre = getenv("ZKHIP_EVALH_PREADD");  // "0" keeps every sum an instruction of its own
    const bool pre_adds = !(env_pre && *env_pre == '0');
    ShapeHash sh;
    sh.word(((uint64_t)desc->basis << 32) | lds_budget | ((uint64_t)do_fuse << 31) | ((uint64_t)pre_adds << 30));
    if (desc->basis != ZKHIP_BASIS_LAGRANGE) sh.word(((uint64_t)desc->term_select << 32) | (desc->term_select ? desc->term_degree_max : 0));
    sh.word(((uint64_t)desc->num_fixed << 32) | desc->num_advice);
    sh.word(((uint64_t)desc->num_instance << 32) | desc->num_challenges);
    sh.word(desc->custom_gates ? 1 : 0);
    if (desc->custom_gates) sh.graph(desc->custom_gates);
    sh.word(desc->num_lookups);
    for (uint32_t i = 0; i < desc->num_lookups; i++) sh.graph(&desc->lookups[i].graph);
    sh.word(((uint64_t)desc->num_perm_sets << 32) | desc->num_perm_columns);
    if (desc->num_perm_sets) {
        sh.word(((uint64_t)desc->perm_chunk_len << 32) | desc->blinding_factors);
        sh.bytes(desc->perm_column_kind, (size_t)desc->num_perm_columns * 4);
        sh.bytes(desc->perm_column_index, (size_t)desc->num_perm_columns * 4);
    }
    const std::pair<uint64_t, uint64_t> key(sh.a, sh.b);
    auto it = ctx->evalh_cache.find(key);
    if (it != ctx->evalh_cache.end()) {
        prog = it->second;
        ctx->evalh_cache_hits++;
        return ZKHIP_OK;
    }
    ctx->evalh_builds++;
    prog = std::make_shared<EvalhProgram>();
    int rc = build_program(ctx, desc, lagrange, lds_budget, do_fuse, pre_adds, *prog);
    if (rc) return rc;
    if (ctx->evalh_cache.size() >= 4096) ctx->evalh_cache.clear();
    ctx->evalh_cache[key] = prog;
    return ZKHIP_OK;
}

// what every use of a descriptor's shape fields checks before the program is built or looked up
static int check_shape(zkhip_ctx* ctx, const zkhip_evalh_desc* desc) {
    ZK_CHECK(ctx, desc->basis <= 1, "evaluate_h: basis %u", desc->basis);
    ZK_CHECK(ctx, desc->term_select <= 2, "evaluate_h: term_select %u", desc->term_select);
    ZK_CHECK(ctx, !desc->num_lookups || desc->lookups, "evaluate_h: lookups missing");
    if (desc->num_perm_sets) {
        ZK_CHECK(ctx, desc->perm_chunk_len >= 1, "evaluate_h: perm_chunk_len = 0");
        ZK_CHECK(ctx, desc->perm_column_kind && desc->perm_column_index, "evaluate_h: permutation arrays missing");
        ZK_CHECK(ctx, (desc->num_perm_columns + desc->perm_chunk_len - 1) / desc->perm_chunk_len == desc->num_perm_sets,
                 "evaluate_h: %u permutation columns in chunks of %u do not make %u sets", desc->num_perm_columns,
                 desc->perm_chunk_len, desc->num_perm_sets);
    }
    if (desc->custom_gates) ZK_CHECK(ctx, graph_arrays_ok(desc->custom_gates), "evaluate_h: graph arrays missing");
    for (uint32_t i = 0; i < desc->num_lookups; i++)
        ZK_CHECK(ctx, graph_arrays_ok(&desc->lookups[i].graph), "evaluate_h: graph arrays missing");
    if (desc->basis == ZKHIP_BASIS_LAGRANGE)
        ZK_CHECK(ctx, desc->custom_gates && !desc->num_lookups && !desc->num_perm_sets && !desc->num_cosets &&
                          desc->poly_form == ZKHIP_FORM_EXTENDED,
                 "evaluate_h: the Lagrange basis takes one graph, columns as given, no lookup / permutation / coset arguments");
    return ZKHIP_OK;
}

// The scalar table of one call, appended to `scal` in the interpreter's 2^261 Montgomery domain (x R -> 32 x R): beta,
// gamma, theta, y, one, the challenges, `extra` (beta * delta^c per permutation column), then the program's term weights
// y^(n_terms - 1 - i) and the y^gap of its groups.
static int call_scalars(zkhip_ctx* ctx, const zkhip_evalh_desc* desc, const EvalhProgram& prog, const std::vector<uint64_t>& extra,
                        std::vector<uint64_t>& scal) {
    const size_t at = scal.size();
    scal.resize(at + (size_t)(SC_CHALLENGE0 + desc->num_challenges) * 4);
    uint64_t* s = &scal[at];
    memcpy(&s[4 * SC_BETA], desc->beta, 32);
    memcpy(&s[4 * SC_GAMMA], desc->gamma, 32);
    memcpy(&s[4 * SC_THETA], desc->theta, 32);
    memcpy(&s[4 * SC_Y], desc->y, 32);
    memcpy(&s[4 * SC_ONE], HFR.r, 32);
    if (desc->num_challenges) memcpy(&s[4 * SC_CHALLENGE0], desc->challenges, (size_t)desc->num_challenges * 32);
    scal.insert(scal.end(), extra.begin(), extra.end());
    if (prog.n_terms) {
        ZK_CHECK(ctx, scal.size() - at == (size_t)prog.sc_y0 * 4, "evaluate_h: internal error, scalar table mismatch");
        scal.resize(at + ((size_t)prog.sc_y0 + prog.n_terms) * 4);
        uint64_t cur[4], y[4];
        memcpy(cur, HFR.r, 32);
        memcpy(y, desc->y, 32);
        for (uint32_t i = prog.n_terms; i-- > 0;) {      // term i weighs y^(n_terms - 1 - i)
            memcpy(&scal[at + ((size_t)prog.sc_y0 + i) * 4], cur, 32);
            h_mul(HFR, cur, cur, y);
        }
        for (uint32_t g : prog.gaps) {
            h_pow(HFR, cur, y, g);
            scal.insert(scal.end(), cur, cur + 4);
        }
    }
    uint64_t k32[4];
    memcpy(k32, FrParams::TWO5, 32);
    for (size_t i = at; i + 4 <= scal.size(); i += 4) h_mul(HFR, &scal[i], &scal[i], k32);
    return ZKHIP_OK;
}

// size of call_scalars' table for this program, in field elements
static size_t call_scalar_count(const zkhip_evalh_desc* desc, const EvalhProgram& prog, size_t extra_elements) {
    return (size_t)SC_CHALLENGE0 + desc->num_challenges + extra_elements + (prog.n_terms ? prog.n_terms + prog.gaps.size() : 0);
}

// Link: operand words -> (header, device pointer).  `linked`: n_instr * INS_WORDS words; res_link: the result operand.
static int link_program(zkhip_ctx* ctx, const EvalhProgram& prog, const std::vector<const fe*>& cols, int32_t rot_scale,
                        const fe* d_scal, const fe* d_consts, uint32_t* linked, uint32_t res_link[3]) {
    for (int32_t r : prog.rots)
        ZK_CHECK(ctx, std::abs((int64_t)r) * rot_scale < (1 << 26), "evaluate_h: rotation %d out of range", r);
    auto link = [&](uint32_t w, uint32_t* o) {
        const uint32_t kind = w >> 28, pl = w & 0x0fffffffu;
        uint64_t ptr = 0;
        uint32_t hdr;
        switch (kind) {
            case K_CONST: hdr = L_UNI << 28; ptr = (uint64_t)(uintptr_t)(d_consts + pl); break;
            case K_SCALAR: hdr = L_UNI << 28; ptr = (uint64_t)(uintptr_t)(d_scal + pl); break;
            case K_COL: {
                const int32_t delta = prog.rots[pl & 0xfff] * rot_scale;
                hdr = (L_ROW << 28) | ((uint32_t)delta & 0x0fffffffu);
                ptr = (uint64_t)(uintptr_t)cols[pl >> 12];
                break;
            }
            case K_SLOT: hdr = (L_SLOT << 28) | pl; break;
            case K_ACC: hdr = L_ACC << 28; break;
            default: hdr = L_ZERO << 28; break;
        }
        o[0] = hdr;
        o[1] = (uint32_t)ptr;
        o[2] = (uint32_t)(ptr >> 32);
    };
    const size_t n_instr = prog.words.size() / IW;
    memset(linked, 0, n_instr * INS_WORDS * 4);
    for (size_t pc = 0; pc < n_instr; pc++) {
        uint32_t* o = &linked[pc * INS_WORDS];
        const uint32_t* w = &prog.words[pc * IW];
        o[0] = w[0] | ((w[1] == DST_ACC_V ? DST_ACC : w[1]) << DST_SHIFT);
        link(w[2], o + 1);
        link(w[3], o + 4);
        link(w[4], o + 7);
        link(w[5], o + 10);
    }
    link(prog.result, res_link);
    return ZKHIP_OK;
}

static int launch_program(zkhip_ctx* ctx, const EvalhProgram& prog, const uint32_t* d_prog, const uint32_t res_link[3],
                          int32_t* d_spill, uint32_t log_rows, size_t row_first, size_t rows, fe* d_out) {
    InterpArgs a;
    memset(&a, 0, sizeof a);
    a.n_instr = (uint32_t)(prog.words.size() / IW);
    a.result_hdr = res_link[0];
    a.lds_slots = prog.lds_slots;
    a.spill = prog.n_slots > prog.lds_slots ? d_spill : nullptr;
    a.log_size = log_rows;
    a.row_first = (uint32_t)row_first;
    a.rows = (uint32_t)rows;
    a.out = d_out;
    const size_t lds_bytes = (size_t)(prog.lds_slots > 1 ? prog.lds_slots - 1 : 1) * EV_THREADS * 36;
    if (!ctx->evalh_attr_set) {
        ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(evalh_interp_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->evalh_attr_set = true;
    }
    TimedRegion tr(ctx, "evalh");
    hipLaunchKernelGGL(evalh_interp_kernel, dim3((unsigned)((rows + EV_THREADS - 1) / EV_THREADS)), dim3(EV_THREADS), lds_bytes,
                       ctx->stream, d_prog, a);
    ZK_HIP(ctx, hipGetLastError());
    return ZKHIP_OK;
}

void evalh_drop_ctx(zkhip_ctx* ctx) {
    ctx->evalh_cache.clear();
    for (int i = 0; i < 4; i++) {
        if (ctx->evalh_buf[i]) (void)hipFree(ctx->evalh_buf[i]);
        ctx->evalh_buf[i] = nullptr;
        ctx->evalh_buf_bytes[i] = 0;
    }
}

}  // namespace zkhip

extern "C" int zkhip_evaluate_h_prepare(zkhip_domain* dom, const zkhip_evalh_desc* desc) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, desc, "evaluate_h_prepare: null descriptor");
    int rc = check_shape(ctx, desc);
    if (rc) return rc;
    std::shared_ptr<EvalhProgram> prog;
    rc = get_program(ctx, desc, desc->basis == ZKHIP_BASIS_LAGRANGE, prog);
    if (!rc) memcpy(ctx->evalh_stats, prog->stats, sizeof ctx->evalh_stats);   // zkhip_evaluate_h_last_stats: of this program
    return rc;
}

extern "C" int zkhip_evaluate_h_columns_used(zkhip_domain* dom, const zkhip_evalh_desc* desc, uint8_t* used, uint32_t capacity,
                                             uint32_t* count) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, desc && used && count, "evaluate_h_columns_used: null pointer");
    int rc = check_shape(ctx, desc);
    if (rc) return rc;
    std::shared_ptr<EvalhProgram> prog;
    if ((rc = get_program(ctx, desc, desc->basis == ZKHIP_BASIS_LAGRANGE, prog))) return rc;
    ZK_CHECK(ctx, prog->col_used.size() <= capacity, "evaluate_h_columns_used: %zu columns, room for %u", prog->col_used.size(),
             capacity);
    memcpy(used, prog->col_used.data(), prog->col_used.size());
    *count = (uint32_t)prog->col_used.size();
    return ZKHIP_OK;
}

extern "C" int zkhip_evaluate_h(zkhip_domain* dom, const zkhip_evalh_desc* desc, uint64_t* out) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, desc && out, "evaluate_h: null descriptor or output");
    ZK_CHECK(ctx, desc->mem <= 1 && desc->poly_form <= 1, "evaluate_h: bad mem / poly_form");
    int rc = check_shape(ctx, desc);
    if (rc) return rc;
    // Lagrange basis: a graph evaluated over the 2^k rows of the circuit itself (rotations move single rows), what the
    // prover does with the lookup expressions before it sorts them
    const bool lagrange = desc->basis == ZKHIP_BASIS_LAGRANGE;
    if (!lagrange) ZK_CHECK(ctx, desc->l0 && desc->l_last && desc->l_active_row, "evaluate_h: l0 / l_last / l_active_row missing");
    ZK_CHECK(ctx, !desc->num_fixed || desc->fixed_cosets, "evaluate_h: fixed_cosets missing");
    ZK_CHECK(ctx, !desc->num_advice || desc->advice, "evaluate_h: advice missing");
    ZK_CHECK(ctx, !desc->num_instance || desc->instance, "evaluate_h: instance missing");
    ZK_CHECK(ctx, !desc->num_challenges || desc->challenges, "evaluate_h: challenges missing");
    if (desc->num_perm_sets)
        ZK_CHECK(ctx, desc->perm_sigma_cosets && desc->perm_product_cosets, "evaluate_h: permutation arrays missing");
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t ek = dom->extended_k, k = dom->k;
    const size_t n = (size_t)1 << k;
    const bool host = desc->mem == ZKHIP_MEM_HOST;
    const bool coeff = desc->poly_form == ZKHIP_FORM_COEFF;
    Arena ar{ctx, {}};
    // Rows evaluated: the whole extended domain, or a subset of its cosets in compact layout.  A rotation by r moves
    // r * 2^e extended rows and so stays inside a coset; in the compact layout it moves r * num_cosets rows.
    const bool compact = desc->num_cosets != 0;
    CosetList cl;
    cl.num = 1u << (ek - k);
    if (compact && (rc = coset_list_check(dom, desc->cosets, desc->num_cosets, &cl))) return rc;
    if (lagrange) cl.num = 1;
    uint32_t log_rows = k;
    while ((1u << (log_rows - k)) < cl.num) log_rows++;
    const size_t size = (size_t)1 << log_rows;
    // Rows evaluated: all `size` rows, or the window [row_first, row_first + row_count) of them (a rank that shares a
    // coset with other ranks evaluates a contiguous part of its rows; the columns are still whole, so rotations need no
    // exchange of neighbouring rows).  `out` then holds row_count rows.
    const size_t row_first = desc->row_count ? desc->row_first : 0;
    const size_t rows = desc->row_count ? desc->row_count : size;
    ZK_CHECK(ctx, row_first + rows <= size, "evaluate_h: rows [%zu, %zu) of %zu", row_first, row_first + rows, size);

    // ---- the program of this shape: from the context's cache, or built now --------------------------------------
    std::shared_ptr<EvalhProgram> prog;
    if ((rc = get_program(ctx, desc, lagrange, prog))) return rc;
    memcpy(ctx->evalh_stats, prog->stats, sizeof ctx->evalh_stats);

    // Bring one polynomial to an extended coset on the device.  `ext_in`: already extended.
    fe *tmp0 = nullptr, *tmp1 = nullptr;
    if (coeff) {
        rc = ensure_scratch(ctx, (compact ? domain_coeff_to_cosets_tmp(dom, cl, 1) : 2 * size) * 32);
        if (rc) return rc;
        tmp0 = (fe*)ctx->scratch;
        tmp1 = tmp0 + size;
    }
    auto resolve = [&](const uint64_t* p, bool ext_in, const fe** outp) -> int {
        ZK_CHECK(ctx, p != nullptr, "evaluate_h: null polynomial pointer");
        if (ext_in) {
            if (!host) { *outp = (const fe*)p; return (int)ZKHIP_OK; }
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
        r = compact ? domain_coeff_to_cosets(dom, d_coeff, 1, cl, (fe*)d_ext, tmp0)
                    : domain_coeff_to_extended(dom, d_coeff, (fe*)d_ext, 1, tmp0, tmp1);
        if (r) return r;
        *outp = (const fe*)d_ext;
        return ZKHIP_OK;
    };

    // ---- unified column table: fixed | advice | instance | l0 l_last l_active | X | perm z | perm sigma | lookups ----
    const ColLayout L(desc, lagrange);
    std::vector<const fe*> cols;
    cols.reserve(L.total);
    auto push = [&](const uint64_t* p, bool ext_in) -> int {
        const fe* d;
        int r = resolve(p, ext_in, &d);
        if (r) return r;
        cols.push_back(d);
        return ZKHIP_OK;
    };
    for (uint32_t i = 0; i < desc->num_fixed; i++)
        if ((rc = push(desc->fixed_cosets[i], true))) return rc;
    for (uint32_t i = 0; i < desc->num_advice; i++)
        if ((rc = push(desc->advice[i], !coeff))) return rc;
    for (uint32_t i = 0; i < desc->num_instance; i++)
        if ((rc = push(desc->instance[i], !coeff))) return rc;
    if (!lagrange && ((rc = push(desc->l0, true)) || (rc = push(desc->l_last, true)) || (rc = push(desc->l_active_row, true))))
        return rc;
    std::vector<uint64_t> extra_scalars;   // after the challenges: beta * delta^c per permutation column
    if (desc->num_perm_sets) {
        const fe* d_x;
        if (compact) {
            void* xp;
            if ((rc = ar.alloc(size * 32, &xp)) || (rc = domain_coset_xpoints(dom, cl, (fe*)xp))) return rc;
            d_x = (const fe*)xp;
        } else {
            if (!dom->d_xpoints) {
                ZK_HIP(ctx, hipMalloc(&dom->d_xpoints, size * 32));
                hipLaunchKernelGGL(xpoints_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (fe*)dom->d_xpoints, (const fe*)dom->plan_ext_fwd->d_lo, (const fe*)dom->plan_ext_fwd->d_hi,
                                   dom->plan_ext_fwd->h, (uint32_t)size);
                ZK_HIP(ctx, hipGetLastError());
            }
            d_x = (const fe*)dom->d_xpoints;
        }
        cols.push_back(d_x);
        for (uint32_t s = 0; s < desc->num_perm_sets; s++)
            if ((rc = push(desc->perm_product_cosets[s], true))) return rc;
        for (uint32_t c = 0; c < desc->num_perm_columns; c++)
            if ((rc = push(desc->perm_sigma_cosets[c], true))) return rc;
        uint64_t cur[4], delta[4];
        memcpy(delta, FrParams::DELTA, 32);
        memcpy(cur, desc->beta, 32);
        for (uint32_t c = 0; c < desc->num_perm_columns; c++) {
            extra_scalars.insert(extra_scalars.end(), cur, cur + 4);
            h_mul(HFR, cur, cur, delta);
        }
    }
    for (uint32_t i = 0; i < desc->num_lookups; i++) {
        const zkhip_lookup& lk = desc->lookups[i];
        if ((rc = push(lk.product, !coeff)) || (rc = push(lk.permuted_input, !coeff)) || (rc = push(lk.permuted_table, !coeff)))
            return rc;
    }
    ZK_CHECK(ctx, cols.size() == L.total && prog->n_cols == L.total, "evaluate_h: internal error, column table mismatch");

    // ---- scalars of this call ------------------------------------------------------------------------------------
    std::vector<uint64_t> scal;
    if ((rc = call_scalars(ctx, desc, *prog, extra_scalars, scal))) return rc;

    const size_t n_instr = prog->words.size() / IW;
    void*& d_prog_v = ctx->evalh_buf[0];
    void*& d_scal_v = ctx->evalh_buf[1];
    void*& d_consts_v = ctx->evalh_buf[2];
    void*& d_spill_v = ctx->evalh_buf[3];
    if ((rc = grow(ctx, d_prog_v, ctx->evalh_buf_bytes[0], std::max<size_t>(1, n_instr) * INS_WORDS * 4)) ||
        (rc = grow(ctx, d_scal_v, ctx->evalh_buf_bytes[1], scal.size() * 8)) ||
        (rc = grow(ctx, d_consts_v, ctx->evalh_buf_bytes[2], std::max<size_t>(32, prog->consts.size() * 8))))
        return rc;
    if (prog->n_slots > prog->lds_slots &&
        (rc = grow(ctx, d_spill_v, ctx->evalh_buf_bytes[3], (size_t)(prog->n_slots - prog->lds_slots) * rows * 36)))
        return rc;
    fe* d_scal = (fe*)d_scal_v;
    fe* d_consts = (fe*)d_consts_v;
    ZK_HIP(ctx, hipMemcpyAsync(d_scal, scal.data(), scal.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (!prog->consts.empty())
        ZK_HIP(ctx, hipMemcpyAsync(d_consts, prog->consts.data(), prog->consts.size() * 8, hipMemcpyHostToDevice, ctx->stream));

    std::vector<uint32_t>& linked = ctx->evalh_linked;
    linked.resize(std::max<size_t>(1, n_instr * INS_WORDS));
    uint32_t res_link[3];
    if ((rc = link_program(ctx, *prog, cols, (int32_t)cl.num, d_scal, d_consts, linked.data(), res_link))) return rc;
    if (n_instr) ZK_HIP(ctx, hipMemcpyAsync(d_prog_v, linked.data(), n_instr * INS_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
    fe* d_out;
    if (host) {
        void* p;
        if ((rc = ar.alloc(rows * 32, &p))) return rc;
        d_out = (fe*)p;
    } else {
        d_out = (fe*)out;
    }
    if ((rc = launch_program(ctx, *prog, (const uint32_t*)d_prog_v, res_link, (int32_t*)d_spill_v, log_rows, row_first, rows, d_out)))
        return rc;
    if (host) ZK_HIP(ctx, hipMemcpyAsync(out, d_out, rows * 32, hipMemcpyDeviceToHost, ctx->stream));
    // the scalars, the linked program and the spill buffer belong to the context and are rewritten by the next call
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKHIP_OK;
}

extern "C" int zkhip_evaluate_graphs_dev(zkhip_domain* dom, const zkhip_evalh_desc* desc, const zkhip_graph* graphs, uint32_t count,
                                         uint64_t* const* d_outs) {
    if (!dom) return ZKHIP_ERR_INVALID;
    zkhip_ctx* ctx = dom->ctx;
    ZK_CHECK(ctx, desc && (count == 0 || (graphs && d_outs)), "evaluate_graphs: null pointer");
    ZK_CHECK(ctx, desc->basis == ZKHIP_BASIS_LAGRANGE && desc->mem == ZKHIP_MEM_DEVICE && desc->poly_form == ZKHIP_FORM_EXTENDED &&
                      !desc->row_count,
             "evaluate_graphs: the graphs run over the 2^k rows of device-resident Lagrange-basis columns");
    ZK_CHECK(ctx, !desc->num_fixed || desc->fixed_cosets, "evaluate_graphs: fixed columns missing");
    ZK_CHECK(ctx, !desc->num_advice || desc->advice, "evaluate_graphs: advice columns missing");
    ZK_CHECK(ctx, !desc->num_instance || desc->instance, "evaluate_graphs: instance columns missing");
    ZK_CHECK(ctx, !desc->num_challenges || desc->challenges, "evaluate_graphs: challenges missing");
    if (!count) return ZKHIP_OK;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t k = dom->k;
    const size_t n = (size_t)1 << k;
    std::vector<const fe*> cols;
    cols.reserve((size_t)desc->num_fixed + desc->num_advice + desc->num_instance);
    auto push = [&](const uint64_t* const* ps, uint32_t cnt) -> int {
        for (uint32_t i = 0; i < cnt; i++) {
            ZK_CHECK(ctx, ps[i] != nullptr, "evaluate_graphs: null column pointer");
            cols.push_back((const fe*)ps[i]);
        }
        return (int)ZKHIP_OK;
    };
    int rc;
    if ((rc = push(desc->fixed_cosets, desc->num_fixed)) || (rc = push(desc->advice, desc->num_advice)) ||
        (rc = push(desc->instance, desc->num_instance)))
        return rc;
    // the programs (from the context's cache after the first proof) and where each one's words, scalars and constants go
    std::vector<std::shared_ptr<EvalhProgram>> progs(count);
    std::vector<size_t> at_words(count + 1, 0), at_scal(count + 1, 0), at_consts(count + 1, 0);
    size_t spill_bytes = 0;
    zkhip_evalh_desc one = *desc;
    one.num_lookups = one.num_perm_sets = one.num_perm_columns = one.num_cosets = 0;
    const std::vector<uint64_t> no_extra;
    for (uint32_t i = 0; i < count; i++) {
        ZK_CHECK(ctx, d_outs[i] != nullptr, "evaluate_graphs: null output pointer");
        one.custom_gates = &graphs[i];
        if ((rc = check_shape(ctx, &one)) || (rc = get_program(ctx, &one, true, progs[i]))) return rc;
        const EvalhProgram& p = *progs[i];
        ZK_CHECK(ctx, p.n_cols == cols.size(), "evaluate_graphs: internal error, column table mismatch");
        at_words[i + 1] = at_words[i] + (p.words.size() / IW) * INS_WORDS;
        at_scal[i + 1] = at_scal[i] + call_scalar_count(&one, p, 0) * 4;
        at_consts[i + 1] = at_consts[i] + p.consts.size();
        if (p.n_slots > p.lds_slots) spill_bytes = std::max(spill_bytes, (size_t)(p.n_slots - p.lds_slots) * n * 36);
    }
    if ((rc = grow(ctx, ctx->evalh_buf[0], ctx->evalh_buf_bytes[0], std::max<size_t>(1, at_words[count]) * 4)) ||
        (rc = grow(ctx, ctx->evalh_buf[1], ctx->evalh_buf_bytes[1], at_scal[count] * 8)) ||
        (rc = grow(ctx, ctx->evalh_buf[2], ctx->evalh_buf_bytes[2], std::max<size_t>(32, at_consts[count] * 8))) ||
        (spill_bytes && (rc = grow(ctx, ctx->evalh_buf[3], ctx->evalh_buf_bytes[3], spill_bytes))))
        return rc;
    uint32_t* d_words = (uint32_t*)ctx->evalh_buf[0];
    uint64_t* d_scal = (uint64_t*)ctx->evalh_buf[1];
    uint64_t* d_consts = (uint64_t*)ctx->evalh_buf[2];
    // one staging pass on the host, three copies, then the launches back to back: the kernels of one stream run in order,
    // so they share the spill buffer, and nothing waits for the host between them
    std::vector<uint32_t>& linked = ctx->evalh_linked;
    linked.resize(std::max<size_t>(1, at_words[count]));
    std::vector<uint64_t> scal, consts;
    scal.reserve(at_scal[count]);
    consts.reserve(at_consts[count]);
    std::vector<uint32_t> res_links((size_t)count * 3);
    for (uint32_t i = 0; i < count; i++) {
        const EvalhProgram& p = *progs[i];
        one.custom_gates = &graphs[i];
        if ((rc = call_scalars(ctx, &one, p, no_extra, scal))) return rc;
        ZK_CHECK(ctx, scal.size() == at_scal[i + 1], "evaluate_graphs: internal error, scalar table size");
        consts.insert(consts.end(), p.consts.begin(), p.consts.end());
        if ((rc = link_program(ctx, p, cols, 1, (const fe*)(d_scal + at_scal[i]), (const fe*)(d_consts + at_consts[i]),
                               &linked[at_words[i]], &res_links[(size_t)i * 3])))
            return rc;
    }
    ZK_HIP(ctx, hipMemcpyAsync(d_scal, scal.data(), scal.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (!consts.empty()) ZK_HIP(ctx, hipMemcpyAsync(d_consts, consts.data(), consts.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (at_words[count])
        ZK_HIP(ctx, hipMemcpyAsync(d_words, linked.data(), at_words[count] * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!spill_bytes && count <= 65535) {
        // no program spills value slots (they would share the spill buffer): ONE launch, blockIdx.y selects the program
        std::vector<ProgEntry> table(count);
        uint32_t lds_slots = 1;
        for (uint32_t i = 0; i < count; i++) {
            table[i] = {(uint32_t)at_words[i], (uint32_t)(progs[i]->words.size() / IW), res_links[(size_t)i * 3], progs[i]->lds_slots,
                        (fe*)d_outs[i], 0};
            lds_slots = std::max(lds_slots, progs[i]->lds_slots);
        }
        if ((rc = grow(ctx, ctx->evalh_buf[3], ctx->evalh_buf_bytes[3], table.size() * sizeof(ProgEntry)))) return rc;
        ZK_HIP(ctx, hipMemcpyAsync(ctx->evalh_buf[3], table.data(), table.size() * sizeof(ProgEntry), hipMemcpyHostToDevice, ctx->stream));
        InterpArgs a;
        memset(&a, 0, sizeof a);
        a.table = (const ProgEntry*)ctx->evalh_buf[3];
        a.log_size = k;
        a.rows = (uint32_t)n;
        const size_t lds_bytes = (size_t)(lds_slots > 1 ? lds_slots - 1 : 1) * EV_THREADS * 36;
        if (!ctx->evalh_attr_set) {
            ZK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(evalh_interp_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            ctx->evalh_attr_set = true;
        }
        {
            TimedRegion tr(ctx, "evalh");
            hipLaunchKernelGGL(evalh_interp_kernel, dim3((unsigned)((n + EV_THREADS - 1) / EV_THREADS), count), dim3(EV_THREADS),
                               lds_bytes, ctx->stream, (const uint32_t*)d_words, a);
            ZK_HIP(ctx, hipGetLastError());
        }
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // `table` is this call's own memory
        return ZKHIP_OK;
    }
    for (uint32_t i = 0; i < count; i++)
        if ((rc = launch_program(ctx, *progs[i], d_words + at_words[i], &res_links[(size_t)i * 3], (int32_t*)ctx->evalh_buf[3], k, 0,
                                 n, (fe*)d_outs[i])))
            break;
    // the staged tables belong to the context and are rewritten by the next call
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

extern "C" int zkhip_evaluate_h_cache_stats(zkhip_ctx* ctx, uint64_t out[2]) {
    if (!ctx || !out) return ZKHIP_ERR_INVALID;
    out[0] = ctx->evalh_builds;
    out[1] = ctx->evalh_cache_hits;
    return ZKHIP_OK;
}

extern "C" int zkhip_evaluate_h_last_stats(zkhip_ctx* ctx, uint64_t out[4]) {
    if (!ctx || !out) return ZKHIP_ERR_INVALID;
    memcpy(out, ctx->evalh_stats, sizeof ctx->evalh_stats);
    return ZKHIP_OK;
}
