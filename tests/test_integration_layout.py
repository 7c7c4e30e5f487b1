"""INTEGRATION.md section 2 shows the Rust `-sys` declarations a maintainer of the reference would add.  No Rust
toolchain exists in the image, so nothing compiles them: this test ties them to include/probly_search_amd.h
mechanically.  Every `#[repr(C)] pub struct` of the document is laid out by the repr(C) rules (field order,
primitive widths, natural alignment) and compared with what the C compiler says about the header's struct of the
same name - sizeof, and offsetof of every field, by name.  The field NAME lists must agree too, so a field added on
one side only fails here.  (src/score/calculator.rs:9-70, src/query.rs:10-15 are what the structs mirror.)"""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIM = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "c_int": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8, "isize": 8}


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    s = s.replace("->", "\u2192")  # (the arrow of a fn type is not a closing bracket)
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out if x.strip()]


def rust_structs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    sec = re.sub(r"//[^\n]*", "", sec)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+)\s*\{", sec):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth:
            depth += {"{": 1, "}": -1}.get(sec[j], 0)
            j += 1
        fields = []
        for f in split_top(sec[i:j - 1]):
            fm = re.match(r"(?:pub\s+)?(\w+)\s*:\s*(.+)$", f, re.S)
            assert fm, (name, f)
            fields.append((fm.group(1), " ".join(fm.group(2).split())))
        structs[name] = fields
    return structs


def layout(structs, name, seen=()):
    """-> (size, align, [(field, offset)]) by the repr(C) rules."""
    assert name not in seen
    off, align, out = 0, 1, []
    for fname, ty in structs[name]:
        if ty.startswith(("*const", "*mut", "Option<unsafe extern", "Option<extern")):
            sz = al = 8
        elif ty in PRIM:
            sz = al = PRIM[ty]
        elif ty in structs:
            sz, al, _ = layout(structs, ty, seen + (name,))
        else:
            am = re.match(r"\[(\w+);\s*(\d+)\]", ty)
            assert am and am.group(1) in PRIM, (name, fname, ty)
            al = PRIM[am.group(1)]
            sz = al * int(am.group(2))
        off = (off + al - 1) // al * al
        out.append((fname, off))
        off += sz
        align = max(align, al)
    return (off + align - 1) // align * align, align, out


def c_field_names(name):
    h = open(os.path.join(ROOT, "include", "probly_search_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    m = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), h, re.S)
    assert m, "no struct %s in the header" % name
    names = []
    for decl in split_top(m.group(1), ";"):
        fp = re.search(r"\(\s*\*\s*(\w+)\s*\)\s*\(", decl)
        if fp:
            names.append(fp.group(1))
            continue
        for part in split_top(decl):
            names.append(re.search(r"(\w+)\s*(?:\[[^\]]*\])?\s*$", part).group(1))
    return names


def test_rust_declarations_match_the_header(tmp_path):
    structs = {n: f for n, f in rust_structs().items() if f and f[0][0] != "_p"}  # (opaque handles have no layout)
    assert {"ps_str", "ps_result", "ps_term_data", "ps_field_data", "ps_score_callbacks", "ps_scorer_desc", "ps_update_stats",
            "ps_plan_entry"} <= set(structs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "probly_search_amd.h"', "int main(void) {", '  printf("{");']
    for n, fields in structs.items():
        lines.append('  printf("\\"%s\\": {\\"sizeof\\": %%zu", sizeof(%s));' % (n, n))
        for f, _ in fields:
            lines.append('  printf(", \\"%s\\": %%zu", offsetof(%s, %s));' % (f, n, f))
        lines.append('  printf("}, ");')
    lines += ['  printf("\\"_\\": 0}\\n");', "  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = json.loads(subprocess.check_output([str(exe)]))
    for n, fields in structs.items():
        size, _, offs = layout(structs, n)
        assert [f for f, _ in fields] == c_field_names(n), (n, "field names / order differ between INTEGRATION.md and the header")
        assert got[n]["sizeof"] == size, (n, got[n]["sizeof"], size)
        for f, o in offs:
            assert got[n][f] == o, (n, f, got[n][f], o)


def test_the_check_notices_a_drifted_field():
    structs = rust_structs()
    broken = dict(structs)
    broken["ps_result"] = [("key", "u32"), ("score", "f64")]
    assert layout(broken, "ps_result")[0] == 16 and layout(broken, "ps_result")[2] == [("key", 0), ("score", 8)]
    broken["ps_plan_entry"] = [f for f in structs["ps_plan_entry"] if f[0] != "layer"]
    assert layout(broken, "ps_plan_entry")[0] != layout(structs, "ps_plan_entry")[0] or \
        [f for f, _ in broken["ps_plan_entry"]] != c_field_names("ps_plan_entry")
