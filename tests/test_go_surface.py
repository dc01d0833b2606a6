"""The Go decorator (go/servicegraph/graphds.go) cannot be compiled here (no Go toolchain), so its cgo surface is checked
textually: every C.sg_* function, C.SG_* constant and every field of sg_config / sg_stats / sg_event / sg_edge_out it touches
must exist in include/servicegraph.h, with the argument count the header declares; and on the other side it must implement the
eleven methods of the reference's datastore.DataStore (datastore/datastore.go:3-20) with the reference's signatures."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "go", "servicegraph", "graphds.go")
HDR = os.path.join(ROOT, "include", "servicegraph.h")

# datastore/datastore.go:3-20, method name -> (parameter types, result)
DATASTORE = {
    "PersistPod": (["Pod", "string"], "error"),
    "PersistService": (["Service", "string"], "error"),
    "PersistReplicaSet": (["ReplicaSet", "string"], "error"),
    "PersistDeployment": (["Deployment", "string"], "error"),
    "PersistEndpoints": (["Endpoints", "string"], "error"),
    "PersistContainer": (["Container", "string"], "error"),
    "PersistDaemonSet": (["DaemonSet", "string"], "error"),
    "PersistStatefulSet": (["StatefulSet", "string"], "error"),
    "PersistRequest": (["*Request"], "error"),
    "PersistKafkaEvent": (["*KafkaEvent"], "error"),
    "PersistAliveConnection": (["*AliveConnection"], "error"),
}


def _strip_c(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


def _strip_go(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)          # (also drops the cgo preamble comment: it holds no C. references)
    s = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', s)          # string literals may hold anything
    return re.sub(r"//[^\n]*", "", s)


def _header():
    h = _strip_c(open(HDR).read())
    funcs = {}
    for m in re.finditer(r"\b(sg_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = m.group(2).strip()
        funcs[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    consts = set(re.findall(r"#\s*define\s+(SG_[A-Z0-9_]+)", h))
    for body in re.findall(r"\benum\b[^{;]*\{(.*?)\}", h, flags=re.S):
        consts |= set(re.findall(r"\b(SG_[A-Z0-9_]+)\b", body))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(sg_[a-z_]+)\s*\{(.*?)\}\s*\1\s*;", h, flags=re.S):
        fields = set()
        for decl in m.group(2).split(";"):
            for name in re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*(?:,|$)", decl.strip()):
                fields.add(name)
        structs[m.group(1)] = fields
    types = set(structs) | set(re.findall(r"typedef\s+struct\s+[a-z_]+\s*\*?\s*(sg_[a-z_]+)\s*;", h))
    return funcs, consts, structs, types


def _go():
    return _strip_go(open(GO).read())


def test_every_cgo_symbol_the_go_file_names_exists_in_the_header():
    funcs, consts, structs, types = _header()
    g = _go()
    used = set(re.findall(r"\bC\.((?:sg|SG)_[A-Za-z0-9_]+)", g))
    assert len(used) >= 40                                   # (the parse found the file's surface)
    missing = [u for u in sorted(used) if u not in funcs and u not in consts and u not in types]
    assert not missing, f"graphds.go names C symbols include/servicegraph.h does not declare: {missing}"
    # the calls carry as many arguments as the prototypes
    for m in re.finditer(r"\bC\.(sg_[a-z0-9_]+)\(", g):
        name = m.group(1)
        if name not in funcs:
            continue                                         # a type conversion, e.g. C.sg_handle(...)
        depth, i, n, any_arg = 1, m.end(), 0, False
        while depth:
            c = g[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                n += 1
            if depth and not c.isspace():
                any_arg = True
            i += 1
        got = n + 1 if any_arg else 0
        assert got == funcs[name], f"C.{name}: {got} arguments in graphds.go, {funcs[name]} in the header"


def test_every_struct_field_the_go_file_touches_exists_in_the_header():
    _, _, structs, _ = _header()
    g = _go()
    # variables of a C struct type: `var x C.sg_T`, `x C.sg_T` in a parameter list, `var x *C.sg_T` with a view slice built from it
    var_type = {}
    for m in re.finditer(r"\b(?:var\s+)?([a-z][A-Za-z0-9_]*)\s+\*?C\.(sg_config|sg_stats|sg_event|sg_edge_out)\b", g):
        var_type.setdefault(m.group(1), m.group(2))
    assert {"cfg", "st", "ev", "rows"} <= set(var_type), var_type
    # `view := unsafe.Slice(rows, n)` then `r := &view[i]`: r is a *C.sg_edge_out
    for m in re.finditer(r"\b([a-z]\w*)\s*:=\s*unsafe\.Slice\((\w+),", g):
        if m.group(2) in var_type:
            var_type[m.group(1)] = var_type[m.group(2)]
    for m in re.finditer(r"\b([a-z]\w*)\s*(?:,\s*\w+\s*)?:=\s*&(\w+)\[", g):
        if m.group(2) in var_type:
            var_type[m.group(1)] = var_type[m.group(2)]
    assert var_type.get("r") == "sg_edge_out"
    touched = {}
    for var, ty in var_type.items():
        for f in re.findall(r"(?<![\w.])" + re.escape(var) + r"\.([a-z_][a-z0-9_]*)\b", g):
            touched.setdefault(ty, set()).add(f)
    for ty in ("sg_config", "sg_stats", "sg_event", "sg_edge_out"):
        assert touched.get(ty), ty
        unknown = sorted(touched[ty] - structs[ty])
        assert not unknown, f"graphds.go touches {ty}.{unknown}, which include/servicegraph.h does not have"
    # what the rows carry out of the engine is all read
    assert {"from_ref", "to_ref", "count", "err_count", "sum_ns", "max_ns", "sumsq_us", "score", "lat_z", "err_ratio"} <= touched["sg_edge_out"]
    assert {"struct_size", "abi_version", "max_known_nodes", "max_edges", "layers"} <= touched["sg_config"]


def _go_methods():
    out = {}
    for m in re.finditer(r"func\s+\(\s*\w+\s+\*GraphDS\s*\)\s+(\w+)\s*\(([^)]*)\)\s*([^{]*)\{", _go()):
        params = []
        for p in [x.strip() for x in m.group(2).split(",") if x.strip()]:
            parts = p.split()
            params.append(parts[-1].replace("datastore.", ""))
        out[m.group(1)] = (params, m.group(3).strip())
    return out


def test_go_decorator_implements_the_reference_datastore_interface():
    have = _go_methods()
    for name, (params, res) in DATASTORE.items():
        assert name in have, f"GraphDS lacks {name} (datastore/datastore.go:3-20)"
        assert have[name] == (params, res), f"{name}: {have[name]} != {(params, res)}"
    # where the reference is at hand (this container, not the GPU box), the table above is checked against its source
    ref = "/root/reference/datastore/datastore.go"
    if os.path.exists(ref):
        src = _strip_go(open(ref).read())
        body = re.search(r"type\s+DataStore\s+interface\s*\{(.*?)\n\}", src, flags=re.S).group(1)
        found = {}
        for m in re.finditer(r"^\s*(\w+)\(([^)]*)\)\s*(\w+)\s*$", body, flags=re.M):
            found[m.group(1)] = ([p.split()[-1] for p in m.group(2).split(",") if p.strip()], m.group(3))
        assert found == DATASTORE
