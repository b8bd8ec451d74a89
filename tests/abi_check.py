"""Mechanical cross-check of the three declaration layers of the C ABI (test infrastructure, no GPU):

    include/fvad.h  (the contract)  <->  bindings/fvad.zig  (what a Zig host compiles)  <->  formula-vad_amd/binding.py (ctypes)

`parse_c_header` and `parse_zig` reduce both files to the same canonical form -- functions as (return type, [argument
types]), structs as ordered [(field name, type)], constants as {name: value} -- where a type is
(base scalar or struct name, pointer depth, constness per pointer level).  `compare` lists every disagreement.
`c_layout` asks gcc for sizeof / offsetof of every struct in the header; `zig_layout` computes the same numbers from the Zig
declarations by the C ABI's natural-alignment rule (what `extern struct` means).  Nothing here imports the product."""
import re
import subprocess
import tempfile
import os

# ---------------------------------------------------------------- canonical scalar names and sizes (LP64)
C_SCALARS = {"int": "i32", "int32_t": "i32", "int16_t": "i16", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64",
             "size_t": "usize", "float": "f32", "double": "f64", "char": "u8", "void": "void"}
ZIG_SCALARS = {"c_int": "i32", "i32": "i32", "i16": "i16", "u8": "u8", "u32": "u32", "u64": "u64", "usize": "usize",
               "f32": "f32", "f64": "f64", "void": "void", "anyopaque": "void"}
SIZES = {"i32": 4, "i16": 2, "u8": 1, "u32": 4, "u64": 8, "usize": 8, "f32": 4, "f64": 8}
# Zig type names that camel -> snake does not turn into the C name
ZIG_NAME_EXCEPTIONS = {"NSNet2": "fvad_nsnet2", "NSNet2Weights": "fvad_nsnet2_weights"}


def zig_to_c_name(name):
    if name in ZIG_NAME_EXCEPTIONS:
        return ZIG_NAME_EXCEPTIONS[name]
    return "fvad_" + re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()


class Type:
    """base: canonical scalar / C struct name / 'fnptr:<name>'; depth: pointer levels; consts: constness of what each level
    points at, outermost pointer first"""

    def __init__(self, base, depth=0, consts=()):
        self.base, self.depth, self.consts = base, depth, tuple(consts)

    def key(self, with_const=True):
        return (self.base, self.depth, self.consts if with_const else None)

    def __eq__(self, other):
        return self.key() == other.key()

    def __repr__(self):
        return f"{self.base}{'*' * self.depth}{list(self.consts) if self.depth else ''}"


# ---------------------------------------------------------------- C header
def _strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = "\n".join(l for l in text.splitlines() if not l.strip().startswith("#"))
    text = text.replace('extern "C" {', " ")
    return text


def _c_type(spec, declarator_stars):
    """spec: tokens before the declarator ('const float', 'fvad_ctx', 'const fvad_vad_config'); declarator_stars: the
    pointer part of the declarator as a list, innermost first, each True if that pointer is `*const`"""
    toks = spec.split()
    base_const = "const" in toks
    toks = [t for t in toks if t not in ("const", "struct")]
    assert len(toks) == 1, spec
    base = C_SCALARS.get(toks[0], toks[0])
    depth = len(declarator_stars)
    # pointee constness, outermost pointer first: the outermost pointer points at the (depth-1)-th pointer, which is
    # const if ITS star is `*const`; the innermost pointer points at the base
    consts = []
    for level in range(depth - 1, -1, -1):          # level = index of the star that creates this pointer
        consts.append(declarator_stars[level - 1] if level > 0 else base_const)
    return Type(base, depth, consts)


def _split_c_decl(decl):
    """'const float *const *band' -> ('const float', [True, False], 'band');  stars innermost first"""
    decl = decl.strip()
    m = re.match(r"^((?:const\s+|struct\s+)*[A-Za-z_][A-Za-z0-9_]*(?:\s+const)?)\s*(.*)$", decl)
    spec, rest = m.group(1), m.group(2)
    stars = []
    while rest.startswith("*"):
        rest = rest[1:].lstrip()
        if rest.startswith("const") and not re.match(r"const[A-Za-z0-9_]", rest):
            stars.append(True)
            rest = rest[5:].lstrip()
        else:
            stars.append(False)
    name = rest.strip()
    name = re.sub(r"\[.*\]$", "", name)
    return spec, stars, name


def _c_params(arglist):
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return []
    out = []
    for a in arglist.split(","):
        spec, stars, name = _split_c_decl(a)
        out.append((name, _c_type(spec, stars)))
    return out


def parse_c_header(text):
    text = _strip_c(text)
    res = {"functions": {}, "structs": {}, "opaque": set(), "constants": {}, "fnptrs": {}}
    # constants from #define are read from the raw text by the caller; enums here
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", text, flags=re.S):
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            k, v = [x.strip() for x in item.split("=")]
            res["constants"][k] = int(v)
    text = re.sub(r"enum\s*\{.*?\}\s*;", " ", text, flags=re.S)
    # function-pointer typedefs
    for m in re.finditer(r"typedef\s+([A-Za-z_ ]+?)\s*\(\s*\*\s*([A-Za-z_0-9]+)\s*\)\s*\((.*?)\)\s*;", text, flags=re.S):
        spec, stars, _ = _split_c_decl(m.group(1) + " x")
        res["fnptrs"][m.group(2)] = (_c_type(spec, stars), _c_params(m.group(3)))
    text = re.sub(r"typedef\s+[A-Za-z_ ]+?\(\s*\*\s*[A-Za-z_0-9]+\s*\)\s*\(.*?\)\s*;", " ", text, flags=re.S)
    # opaque forward declarations
    for m in re.finditer(r"typedef\s+struct\s+([A-Za-z_0-9]+)\s+([A-Za-z_0-9]+)\s*;", text):
        res["opaque"].add(m.group(2))
    text = re.sub(r"typedef\s+struct\s+[A-Za-z_0-9]+\s+[A-Za-z_0-9]+\s*;", " ", text)
    # struct typedefs
    for m in re.finditer(r"typedef\s+struct\s*(?:[A-Za-z_0-9]+)?\s*\{(.*?)\}\s*([A-Za-z_0-9]+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            first, *more = stmt.split(",")
            spec, stars, name = _split_c_decl(first)
            fields.append((name, _fn_or(spec, stars, res)))
            for extra in more:
                _, stars2, name2 = _split_c_decl("int " + extra.strip())     # dummy spec: only the declarator matters
                fields.append((name2, _fn_or(spec, stars2, res)))
        res["structs"][m.group(2)] = fields
    text = re.sub(r"typedef\s+struct\s*(?:[A-Za-z_0-9]+)?\s*\{.*?\}\s*[A-Za-z_0-9]+\s*;", " ", text, flags=re.S)
    # prototypes
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(fvad_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        spec, stars, _ = _split_c_decl(" ".join(m.group(1).split()) + " x")
        res["functions"][m.group(2)] = (_c_type(spec, stars), _c_params(" ".join(m.group(3).split())))
    return res


def _fn_or(spec, stars, res):
    toks = [t for t in spec.split() if t not in ("const", "struct")]
    if toks and toks[0] in res["fnptrs"]:
        return Type("fnptr:" + toks[0], 0, ())
    return _c_type(spec, stars)


def c_defines(text):
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(FVAD_[A-Z_]+)\s+(-?\d+)\s*$", text, flags=re.M)}


# ---------------------------------------------------------------- Zig binding
def _strip_zig(text):
    return re.sub(r"//[^\n]*", "", text)


def _zig_type(t, fnptr_names=()):
    t = t.strip()
    if t in fnptr_names:
        return Type("fnptr:" + zig_to_c_name(t), 0, ())
    depth, consts = 0, []
    while True:
        t = t.strip()
        if t.startswith("?"):
            t = t[1:]
            continue
        m = re.match(r"^(\*|\[\*(?::0)?\])\s*", t)
        if not m:
            break
        t = t[m.end():]
        is_const = False
        if re.match(r"^const\b", t):
            is_const = True
            t = t[5:]
        depth += 1
        consts.append(is_const)
    t = t.strip()
    base = ZIG_SCALARS.get(t) or zig_to_c_name(t)
    return Type(base, depth, consts)


def _split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def parse_zig(text):
    text = _strip_zig(text)
    res = {"functions": {}, "structs": {}, "opaque": set(), "constants": {}, "fnptrs": {}}
    for m in re.finditer(r"pub const (\w+)\s*=\s*opaque\s*\{\s*\}\s*;", text):
        res["opaque"].add(zig_to_c_name(m.group(1)))
    fnptr_names = []
    for m in re.finditer(r"pub const (\w+)\s*=\s*\?\*const fn\s*\((.*?)\)\s*callconv\(\.C\)\s*(\w+)\s*;", text):
        fnptr_names.append(m.group(1))
        params = []
        for a in _split_top(m.group(2)):
            n, ty = a.split(":", 1)
            params.append((n.strip(), _zig_type(ty)))
        res["fnptrs"][zig_to_c_name(m.group(1))] = (_zig_type(m.group(3)), params)
    for m in re.finditer(r"pub const (\w+)\s*=\s*extern struct\s*\{(.*?)\}\s*;", text, flags=re.S):
        fields = []
        for f in _split_top(m.group(2)):
            f = f.strip()
            if not f:
                continue
            n, ty = f.split(":", 1)
            ty = _split_top(ty, "=")[0]                    # drop the default value
            fields.append((n.strip(), _zig_type(ty, fnptr_names)))
        res["structs"][zig_to_c_name(m.group(1))] = fields
    for m in re.finditer(r'pub extern "c" fn (\w+)\s*\((.*?)\)\s*([^;]+);', text, flags=re.S):
        params = []
        for a in _split_top(m.group(2)):
            if not a.strip():
                continue
            n, ty = a.split(":", 1)
            params.append((n.strip(), _zig_type(ty, fnptr_names)))
        res["functions"][m.group(1)] = (_zig_type(m.group(3), fnptr_names), params)
    m = re.search(r"pub const Status\s*=\s*struct\s*\{(.*?)\}\s*;", text, flags=re.S)
    if m:
        for c in re.finditer(r"pub const (\w+)\s*=\s*(-?\d+)\s*;", m.group(1)):
            name = c.group(1)
            res["constants"]["FVAD_OK" if name == "ok" else "FVAD_" + name.upper()] = int(c.group(2))
    for c in re.finditer(r"^pub const (\w+)\s*=\s*(-?\d+)\s*;", text, flags=re.M):
        res["constants"]["FVAD_" + c.group(1).upper()] = int(c.group(2))
    return res


# ---------------------------------------------------------------- comparison
def compare(c, z, allow_undeclared=()):
    """every disagreement between the header (c) and the Zig binding (z), as a list of strings"""
    bad = []
    for name, (ret, params) in c["functions"].items():
        if name not in z["functions"]:
            if name not in allow_undeclared:
                bad.append(f"function {name}: not declared in the Zig binding")
            continue
        zret, zparams = z["functions"][name]
        if ret != zret:
            bad.append(f"function {name}: returns {ret} in C, {zret} in Zig")
        if len(params) != len(zparams):
            bad.append(f"function {name}: {len(params)} arguments in C, {len(zparams)} in Zig")
            continue
        for i, ((cn, ct), (zn, zt)) in enumerate(zip(params, zparams)):
            if ct != zt:
                bad.append(f"function {name}: argument {i} ({cn}) is {ct} in C, {zt} in Zig")
            if cn != zn:
                bad.append(f"function {name}: argument {i} is named {cn} in C, {zn} in Zig")
    for name in z["functions"]:
        if name not in c["functions"]:
            bad.append(f"function {name}: declared in Zig, not in fvad.h")
    for name, fields in c["structs"].items():
        if name not in z["structs"]:
            bad.append(f"struct {name}: not declared in the Zig binding")
            continue
        zf = z["structs"][name]
        if [n for n, _ in fields] != [n for n, _ in zf]:
            bad.append(f"struct {name}: field order {[n for n, _ in fields]} in C, {[n for n, _ in zf]} in Zig")
            continue
        for (n, ct), (_, zt) in zip(fields, zf):
            if ct != zt:
                bad.append(f"struct {name}.{n}: {ct} in C, {zt} in Zig")
    for name in z["structs"]:
        if name not in c["structs"]:
            bad.append(f"struct {name}: declared in Zig, not in fvad.h")
    for name in c["opaque"] - z["opaque"]:
        bad.append(f"opaque handle {name}: not declared in the Zig binding")
    for name, (ret, params) in c["fnptrs"].items():
        if name not in z["fnptrs"]:
            bad.append(f"callback type {name}: not declared in the Zig binding")
            continue
        zret, zparams = z["fnptrs"][name]
        if ret != zret or [t for _, t in params] != [t for _, t in zparams]:
            bad.append(f"callback type {name}: {ret}({[t for _, t in params]}) in C, {zret}({[t for _, t in zparams]}) in Zig")
    for name, v in c["constants"].items():
        if z["constants"].get(name) != v:
            bad.append(f"constant {name}: {v} in C, {z['constants'].get(name)} in Zig")
    return bad


# ---------------------------------------------------------------- layouts
def c_layout(header_path, structs):
    """{struct: (sizeof, {field: offsetof})} from gcc on the real header"""
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{header_path}"', "int main(void) {"]
    for s, fields in structs.items():
        lines.append(f'printf("S {s} %zu\\n", sizeof({s}));')
        for n, _ in fields:
            lines.append(f'printf("F {s} {n} %zu\\n", offsetof({s}, {n}));')
    lines += ["return 0; }"]
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    res = {}
    for line in out.splitlines():
        p = line.split()
        if p[0] == "S":
            res[p[1]] = (int(p[2]), {})
        else:
            res[p[1]][1][p[2]] = int(p[3])
    return res


def zig_layout(structs):
    """the C ABI layout of the Zig `extern struct`s: natural alignment, nested structs by value"""
    done = {}

    def size_align(t):
        if t.depth > 0 or t.base.startswith("fnptr:"):
            return 8, 8
        if t.base in SIZES:
            return SIZES[t.base], SIZES[t.base]
        return layout(t.base)[0], layout(t.base)[2]

    def layout(name):
        if name in done:
            return done[name]
        off, offs, amax = 0, {}, 1
        for n, t in structs[name]:
            sz, al = size_align(t)
            off = (off + al - 1) // al * al
            offs[n] = off
            off += sz
            amax = max(amax, al)
        done[name] = ((off + amax - 1) // amax * amax, offs, amax)
        return done[name]

    return {s: layout(s)[:2] for s in structs}
