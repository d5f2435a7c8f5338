#!/usr/bin/env python3
"""Static census of the reference SuperCircuit's ConstraintSystem shape.

Reads the reference's Rust sources AS TEXT (nothing from the reference is imported, compiled or executed) and counts, per
sub-circuit, the calls that shape the ConstraintSystem: `meta.advice_column()` / `advice_column_in(..)`, `fixed_column()`,
`lookup_table_column()`, `instance_column()`, `selector()` / `complex_selector()`, `create_gate(..)`, `lookup(..)` /
`lookup_any(..)`, `enable_equality(..)`, with the file:line of every call site.  A call inside `std::array::from_fn`,
`[0; N].map`, `(0..N).map` or a `for _ in A..B` loop counts N times when N resolves to a literal or to a `const` found in
the tree; otherwise the site is listed under `unresolved` (the count is then a lower bound).  Chips from `gadgets/` are
counted once per `configure` call site with the columns their own `configure` creates.

This cannot replace `print_circuit_metrics` (zkevm-circuits/src/super_circuit.rs:462-516, which needs the Rust halo2 fork):
the keccak circuit allocates its cell columns and lookups from a run-time cell manager, and the execution gadgets' merged
lookups are grouped at run time by the columns they touch.  Both are reported as bounded estimates with their derivation.

usage: python tools/supercircuit_census.py [/root/reference] > tests/golden/supercircuit_census.json
"""
import glob
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "zkevm-circuits", "src")

KINDS = [
    ("advice", re.compile(r"\bmeta\.advice_column\(\)|\bcs\.advice_column\(\)")),
    ("advice_second_phase", re.compile(r"\.advice_column_in\(\s*SecondPhase\s*\)")),
    ("advice_other_phase", re.compile(r"\.advice_column_in\(\s*(?!SecondPhase)")),
    ("fixed", re.compile(r"\.fixed_column\(\)")),
    ("table_column", re.compile(r"\.lookup_table_column\(\)")),
    ("instance", re.compile(r"\.instance_column\(\)")),
    ("selector", re.compile(r"\.(?:complex_)?selector\(\)")),
    ("gate", re.compile(r"\.create_gate\(")),
    ("lookup", re.compile(r"\bmeta\.lookup\(|\bcs\.lookup\(")),
    ("lookup_any", re.compile(r"\.lookup_any\(")),
    ("enable_equality", re.compile(r"\.enable_equality\(")),
]

CONST_GENERICS = {"NUM_STATE_HI_COL": 9, "NUM_STATE_LO_COL": 9}   # zkevm-circuits/src/constant.rs:7-12


def rel(path):
    return os.path.relpath(path, REF)


def load(path):
    """source lines up to the file's test module"""
    lines = open(path, encoding="utf-8", errors="replace").read().split("\n")
    for i, ln in enumerate(lines):
        if ln.strip().startswith("#[cfg(test)]"):
            return lines[:i]
    return lines


def strip_comment(ln):
    i = ln.find("//")
    return ln if i < 0 else ln[:i]


def collect_consts():
    out = dict(CONST_GENERICS)
    pat = re.compile(r"\bconst\s+([A-Z][A-Z0-9_]*)\s*:\s*(?:usize|u32|u64|i32)\s*=\s*([^;]+);")
    pending = []
    for root in ("zkevm-circuits/src", "gadgets/src", "poseidon/src", "eth-types/src"):
        for f in glob.glob(os.path.join(REF, root, "**", "*.rs"), recursive=True):
            for ln in open(f, encoding="utf-8", errors="replace"):
                m = pat.search(strip_comment(ln))
                if m:
                    pending.append((m.group(1), m.group(2).strip()))
    for _ in range(4):
        for name, expr in pending:
            if name in out:
                continue
            v = evaluate(expr, out)
            if v is not None:
                out[name] = v
    return out


def evaluate(expr, consts):
    expr = expr.strip().replace("usize", "").replace(" as ", " ")
    expr = re.sub(r"\b(?:[A-Za-z_][A-Za-z0-9_]*::)+([A-Z][A-Z0-9_]*)\b", r"\1", expr)      # Type::CONST -> CONST
    expr = re.sub(r"\b([A-Z][A-Z0-9_]*)\b", lambda m: str(consts[m.group(1)]) if m.group(1) in consts else m.group(0), expr)
    if re.fullmatch(r"[0-9+\-*/() ]+", expr):
        try:
            return int(eval(expr.replace("/", "//"), {"__builtins__": {}}))
        except Exception:
            return None
    return None


CONSTS = None


def statement_window(lines, i):
    """the lines of the statement that contains line i (back to the previous ';', '{' or '}' line end)"""
    j = i
    while j > 0 and j > i - 6:
        prev = strip_comment(lines[j - 1]).rstrip()
        if prev.endswith((";", "{", "}")) or not prev:
            break
        j -= 1
    return " ".join(strip_comment(x) for x in lines[j:i + 1])


def multiplicity(lines, i):
    """(factor, resolved, why) for a call on line i"""
    factor, resolved, why = 1, True, []
    stmt = statement_window(lines, i)
    if "from_fn" in stmt:
        m = re.search(r";\s*([A-Za-z0-9_ +\-*]+)\]\s*=", stmt) or re.search(r";\s*([A-Za-z0-9_ +\-*]+)\]>?\s*=", stmt)
        n = evaluate(m.group(1), CONSTS) if m else None
        if n is None:
            # `let name = from_fn(..)` / `name: from_fn(..)`: the length is in the declaration `name: [..; N]` of the struct
            nm = re.search(r"(?:let\s+(?:mut\s+)?)?([a-z_][a-z0-9_]*)\s*(?::[^=]*)?[=:]\s*(?:std::)?array::from_fn", stmt)
            if nm:
                decl = re.compile(r"\b" + nm.group(1) + r"\s*:\s*\[.*;\s*([A-Za-z0-9_:]+)\s*\]\s*,")
                for other in lines:
                    dm = decl.search(strip_comment(other))
                    if dm:
                        n = evaluate(dm.group(1), CONSTS)
                        if n is not None:
                            break
        if n is None:
            resolved = False
            why.append("array::from_fn of unknown length")
        else:
            factor *= n
            why.append(f"array::from_fn x{n}")
    m = re.search(r"\[\s*(?:0|\(\))\s*;\s*([A-Za-z0-9_ +\-*]+)\]\s*\.map", stmt)
    if m:
        n = evaluate(m.group(1), CONSTS)
        if n is None:
            resolved = False
            why.append(f"[_; {m.group(1).strip()}].map of unknown length")
        else:
            factor *= n
            why.append(f"[_; N].map x{n}")
    m = re.search(r"\((\w+)\.\.(\w+)\)\s*\.map", stmt)
    if m:
        a, b = evaluate(m.group(1), CONSTS), evaluate(m.group(2), CONSTS)
        if a is None or b is None:
            resolved = False
            why.append(f"({m.group(1)}..{m.group(2)}).map of unknown length")
        else:
            factor *= b - a
            why.append(f"range map x{b - a}")
    # enclosing `for` loops: walk back with a brace counter
    depth = 0
    for j in range(i - 1, max(-1, i - 400), -1):
        ln = strip_comment(lines[j])
        depth += ln.count("}") - ln.count("{")
        if depth < 0:                       # line j opens a block that encloses line i
            depth = 0
            m = re.search(r"\bfor\s+.*\s+in\s+(.*)\{\s*$", ln)
            if m:
                rng = m.group(1).strip()
                r = re.fullmatch(r"\(?\s*([A-Za-z0-9_]+)\s*\.\.(=?)\s*([A-Za-z0-9_ +\-*]+?)\s*\)?", rng)
                n = None
                if r:
                    a, b = evaluate(r.group(1), CONSTS), evaluate(r.group(3), CONSTS)
                    if a is not None and b is not None:
                        n = b - a + (1 if r.group(2) else 0)
                if n is None:
                    resolved = False
                    why.append(f"inside `for .. in {rng}` (run-time trip count)")
                else:
                    factor *= n
                    why.append(f"for-loop x{n}")
            if re.search(r"\bfn\s+\w+", ln):
                break
    return factor, resolved, "; ".join(why)


def scan(path, first=None, last=None):
    """every shaping call in a file (optionally a 1-based inclusive line range)"""
    lines = load(path)
    events = []
    for i, raw in enumerate(lines):
        if first is not None and not (first <= i + 1 <= last):
            continue
        ln = strip_comment(raw)
        for kind, pat in KINDS:
            hits = len(pat.findall(ln))
            if not hits:
                continue
            f, ok, why = multiplicity(lines, i)
            events.append({"kind": kind, "at": f"{rel(path)}:{i + 1}", "count": hits * f, "resolved": ok, "how": why})
    return events


def chip_sites(path, chips, first=None, last=None):
    """`Chip::configure(` call sites of gadget chips in a file"""
    lines = load(path)
    out = []
    for i, raw in enumerate(lines):
        if first is not None and not (first <= i + 1 <= last):
            continue
        ln = strip_comment(raw)
        for chip in chips:
            if re.search(r"\b" + chip + r"(?:::<[^>]*>)?::configure\(", ln):
                f, ok, why = multiplicity(lines, i)
                out.append({"chip": chip, "at": f"{rel(path)}:{i + 1}", "count": f, "resolved": ok, "how": why})
    return out


GENERIC_CONFIG = {"BinaryNumberChip": "BinaryNumberConfig", "BinaryNumberWithRealSelectorChip": "BinaryNumberConfig",
                  "LtChip": "LtConfig", "MpiChip": "MpiConfig"}


def generic_sizes(path, config_type, first=None, last=None):
    """the last generic argument of every `field: ConfigType<.., N>` declaration in a file, in order"""
    out = []
    for i, raw in enumerate(load(path)):
        if first is not None and not (first <= i + 1 <= last):
            continue
        m = re.search(r":\s*" + config_type + r"<[^>]*?,\s*([A-Za-z0-9_:]+)\s*>", strip_comment(raw))
        if m:
            out.append((evaluate(m.group(1), CONSTS), m.group(1), f"{rel(path)}:{i + 1}"))
    return out


def summarise(events):
    tot, unresolved = {}, []
    for e in events:
        tot[e["kind"]] = tot.get(e["kind"], 0) + e["count"]
        if not e["resolved"]:
            unresolved.append(e)
    return tot, unresolved


# gadget chips (gadgets/src): what one `configure` creates
GADGET_FILES = {
    "MpiChip": "zkevm-circuits/src/state_circuit/multiple_precision_integer.rs",
    "IsZeroWithRotationConfig": "gadgets/src/is_zero_with_rotation.rs", "IsZeroWithRotationChip": "gadgets/src/is_zero_with_rotation.rs",
    "IsZeroChip": "gadgets/src/is_zero.rs", "IsZeroConfig": "gadgets/src/is_zero.rs",
    "LtChip": "gadgets/src/less_than.rs", "LtConfig": "gadgets/src/less_than.rs",
    "DynamicSelectorChip": "gadgets/src/dynamic_selector.rs", "DynamicSelectorConfig": "gadgets/src/dynamic_selector.rs",
    "BinaryNumberChip": "gadgets/src/binary_number.rs", "BinaryNumberConfig": "gadgets/src/binary_number.rs",
    "BinaryNumberWithRealSelectorChip": "gadgets/src/binary_number_with_real_selector.rs",
    "BinaryNumberWithRealSelectorConfig": "gadgets/src/binary_number_with_real_selector.rs",
}

# table.rs line ranges of each table's `construct` (SURVEY: zkevm-circuits/src/table.rs)
TABLES = {"state": (188, 318), "bytecode": (319, 401), "fixed": (402, 463), "arithmetic": (464, 534), "copy": (535, 638),
          "public": (639, 768), "bitwise": (769, 827), "exp": (828, 871), "keccak": (872, 931), "poseidon": (932, 1000)}

SUBCIRCUITS = {
    # name: (files, table names, only with features, note)
    "core": (["core_circuit.rs", "execution.rs"], [], None),
    "state": (["state_circuit.rs", "state_circuit/ordering.rs"], ["state"], None),
    "bytecode": (["bytecode_circuit.rs"], ["bytecode"], None),
    "copy": (["copy_circuit.rs"], ["copy"], None),
    "public": (["public_circuit.rs"], ["public"], None),
    "arithmetic": (["arithmetic_circuit.rs"], ["arithmetic"], None),
    "bitwise": (["bitwise_circuit.rs"], ["bitwise"], None),
    "exp": (["exp_circuit.rs"], ["exp"], None),
    "fixed": (["fixed_circuit.rs"], ["fixed"], None),
    "keccak": (["keccak_circuit.rs", "keccak_circuit/cell_manager.rs", "keccak_circuit/keccak_packed_multi.rs"], ["keccak"],
               "standard only (feature no_keccak_circuit removes it; fast_test sets that feature)"),
    "poseidon": (["poseidon_circuit.rs"], ["poseidon"], None),
    "super": (["super_circuit.rs", "util.rs"], [], None),
}


def execution_census():
    """what the 78 execution gadgets add to the core circuit: two gates each (execution.rs:2152-2197) and their lookups,
    merged per distinct (table kind, position) signature (execution.rs:2268-2334)"""
    ex = os.path.join(SRC, "execution.rs")
    text = open(ex, encoding="utf-8", errors="replace").read()
    m = re.search(r"macro_rules! get_every_execution_gadgets \{(.*?)\n\}", text, flags=re.S)
    gadgets = re.findall(r"crate::execution::(\w+)::new\(\)", m.group(1)) if m else []
    sigs, per_kind, rotations = {}, {}, set()
    helper = re.compile(r"config\s*\.\s*(get_\w*lookup\w*)\s*\(([^;]*?)\)\s*[;,)]", flags=re.S)
    max_num_row = 0
    for g in gadgets:
        path = os.path.join(SRC, "execution", g + ".rs")
        if not os.path.exists(path):
            continue
        body = "\n".join(strip_comment(x) for x in load(path))
        mm = re.search(r"const NUM_ROW\s*:\s*usize\s*=\s*(\d+)", body)
        if mm:
            max_num_row = max(max_num_row, int(mm.group(1)))
        for r in re.findall(r"Rotation\(\s*(-?\d+)\s*\)", body):
            rotations.add(int(r))
        if "Rotation::prev()" in body:
            rotations.add(-1)
        if "Rotation::next()" in body:
            rotations.add(1)
        for fn, args in helper.findall(body):
            args = re.sub(r"\s+", "", args)
            literal = ",".join(a for a in args.split(",")[1:] if re.fullmatch(r"-?\d+|Rotation\(-?\d+\)|Rotation::\w+\(\)", a))
            key = f"{fn}({literal})"
            sigs.setdefault(key, []).append(g)
            per_kind[fn] = per_kind.get(fn, 0) + 1
    return {"gadgets": len(gadgets), "gadget_list_at": "zkevm-circuits/src/execution.rs:109-191",
            "gates_from_gadgets": 2 * len(gadgets), "gates_at": "zkevm-circuits/src/execution.rs:2152-2197",
            "lookup_helper_calls": sum(per_kind.values()), "lookup_helper_calls_by_helper": per_kind,
            "distinct_lookup_signatures": len(sigs),
            "distinct_lookup_signatures_note": "helper name + its literal position arguments; the reference merges lookups whose "
                                               "entries touch the same columns at the same rotations (execution.rs:2268-2334), so "
                                               "this is the static estimate of the merged lookup_any count of the core circuit",
            "max_NUM_ROW": max_num_row, "literal_rotations": sorted(rotations)}


def main():
    global CONSTS
    CONSTS = collect_consts()
    table_path = os.path.join(SRC, "table.rs")
    gadget_shape = {}
    for chip, f in GADGET_FILES.items():
        ev = scan(os.path.join(REF, f))
        fixed_part, _ = summarise([e for e in ev if e["resolved"]])
        per_n, _ = summarise([e for e in ev if not e["resolved"]])      # sized by the chip's const generic
        gadget_shape[chip] = {"file": f, "creates": fixed_part, "creates_per_generic_size": per_n,
                              "sized_sites": [e["at"] + " " + e["how"] for e in ev if not e["resolved"]]}
    out = {"generated_by": "tools/supercircuit_census.py (static text census of the reference's Rust sources; nothing executed)",
           "constants_used": {k: CONSTS[k] for k in ("NUM_VERS", "NUM_STATE_HI_COL", "NUM_STATE_LO_COL", "NUM_WORDS_TO_ABSORB",
                                                     "NUM_WORDS_TO_SQUEEZE") if k in CONSTS},
           "subcircuits": {}, "gadget_chips": gadget_shape}
    for name, (files, tables, only) in SUBCIRCUITS.items():
        events, chips = [], []
        for f in files:
            p = os.path.join(SRC, f)
            if not os.path.exists(p):
                continue
            events += scan(p)
            chips += chip_sites(p, GADGET_FILES)
        for t in tables:
            a, b = TABLES[t]
            events += scan(table_path, a, b)
            chips += chip_sites(table_path, GADGET_FILES, a, b)
        tot, unres = summarise(events)
        via = {}
        seen = {}
        for c in chips:
            shape = gadget_shape[c["chip"]]
            for kind, n in shape["creates"].items():
                via[kind] = via.get(kind, 0) + n * c["count"]
            if shape["creates_per_generic_size"]:
                cfg = GENERIC_CONFIG.get(c["chip"])
                f = os.path.join(REF, c["at"].split(":")[0])
                rng = None
                if f == table_path:
                    rng = next((TABLES[t] for t in tables if TABLES[t][0] <= int(c["at"].split(":")[1]) <= TABLES[t][1]), None)
                sizes = generic_sizes(f, cfg, *(rng or (None, None))) if cfg else []
                k = seen.get((f, cfg), 0)
                seen[(f, cfg)] = k + 1
                size = sizes[min(k, len(sizes) - 1)] if sizes else (None, "?", "")
                c["generic_size"] = {"value": size[0], "expr": size[1], "declared_at": size[2]}
                if size[0] is None:
                    c["resolved"] = False
                    continue
                for kind, n in shape["creates_per_generic_size"].items():
                    via[kind] = via.get(kind, 0) + n * size[0] * c["count"]
        entry = {"direct": tot, "via_gadget_chips": via, "chip_sites": chips, "sites": events, "unresolved": unres}
        if only:
            entry["present"] = only
        out["subcircuits"][name] = entry
    out["execution_gadgets"] = execution_census()

    def total(kind, skip=()):
        s = 0
        for name, e in out["subcircuits"].items():
            if name in skip:
                continue
            s += e["direct"].get(kind, 0) + e["via_gadget_chips"].get(kind, 0)
        return s

    ex = out["execution_gadgets"]
    # the keccak circuit's cell columns and per-cell lookups come from a run-time cell manager
    # (keccak_circuit/cell_manager.rs:164-180): the figure used is the middle of the survey's bracket
    KECCAK_CELL_COLUMNS = (130, 150)
    KECCAK_LOOKUPS = (80, 120)
    fast_skip = ("keccak", "fixed")
    shapes = {}
    for label, skip, kc, kl in (("standard", (), KECCAK_CELL_COLUMNS, KECCAK_LOOKUPS), ("fast_test", fast_skip, (0, 0), (0, 0))):
        adv = total("advice", skip) + total("advice_second_phase", skip)
        shapes[label] = {
            "advice_counted": adv, "advice_second_phase": total("advice_second_phase", skip),
            "advice_with_keccak_cell_columns": [adv + kc[0], adv + kc[1]],
            "fixed_counted": total("fixed", skip) + total("selector", skip) + total("table_column", skip),
            "instance": total("instance", skip),
            "gates_counted": total("gate", skip) + ex["gates_from_gadgets"],
            "lookups_counted": total("lookup", skip) + total("lookup_any", skip) + ex["distinct_lookup_signatures"],
            "lookups_with_keccak_cell_lookups": [total("lookup", skip) + total("lookup_any", skip) + ex["distinct_lookup_signatures"] + kl[0],
                                                 total("lookup", skip) + total("lookup_any", skip) + ex["distinct_lookup_signatures"] + kl[1]],
            "enable_equality_sites": total("enable_equality", skip),
            "skipped_subcircuits": list(skip),
        }
    shapes["notes"] = {
        "keccak": "cell columns and the lookups made per cell are allocated at run time (zkevm-circuits/src/keccak_circuit/"
                  "cell_manager.rs:164-180, keccak_circuit.rs:376-420); bracket taken from the packed-multi layout at 12 rows "
                  "per round (zkevm-circuits/src/keccak_circuit/param.rs:1-2)",
        "max_rotation": "core gadgets: -max_NUM_ROW .. +1; keccak: (i + 1) * 12 for i < 17 = +204 "
                        "(zkevm-circuits/src/keccak_circuit.rs:158-162)",
        "fast_test": "features k_11, no_fixed_lookup, no_keccak_circuit (zkevm-circuits/Cargo.toml:47-56)",
        "degree": "every core gate and lookup carries q_enable * cnt_is_zero * execution_state_selector (degree 5, "
                  "zkevm-circuits/src/execution.rs:2166,2192,2328): cs.degree() >= 9, extended domain 8n",
    }
    out["shape"] = shapes
    json.dump(out, sys.stdout, indent=1, sort_keys=False)
    print()


if __name__ == "__main__":
    main()
