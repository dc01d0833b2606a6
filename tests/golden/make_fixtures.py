"""Regenerates tests/golden/*.json from the reference checkout (run in the build container only;
/root/reference does not exist on the GPU box, the committed JSON files travel instead).

  pg_kat.json   the Parse / Bind payloads and expected strings of aggregator/pg_test.go
                (TestPostgresParseWithKnownStmt :10-92, TestPostgresParseWithUnknownStmt :94-119).
                The test file is stale (it calls a.parseSqlCommand; the function is
                parsePostgresCommand, aggregator/data.go:1474) but its vectors are valid for it.
  sim_kat.json  the constants of the reference's simulator (main_benchmark_test.go:561-617,
                testconfig/config1.json) restated as one tiny trace with the outputs the reference's
                code produces for it, derived by hand from aggregator/data.go:508-531, 827-870,
                1208-1249, 1740-1767 and datastore/backend.go:824-839.
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def go_string(lit: str) -> bytes:
    """Interpreted Go string literal body -> the bytes []byte(lit) yields (UTF-8)."""
    out = []
    i = 0
    while i < len(lit):
        c = lit[i]
        if c != "\\":
            out.append(c); i += 1; continue
        n = lit[i + 1]
        if n == "u":
            out.append(chr(int(lit[i + 2:i + 6], 16))); i += 6
        elif n == "t":
            out.append("\t"); i += 2
        elif n == "n":
            out.append("\n"); i += 2
        elif n == "r":
            out.append("\r"); i += 2
        elif n in "\"\\":
            out.append(n); i += 2
        else:
            raise ValueError(f"escape \\{n}")
    return "".join(out).encode("utf-8")


def main():
    src = open(os.path.join(REF, "aggregator/pg_test.go"), encoding="utf-8").read()
    lits = re.findall(r'\[\]byte\("((?:[^"\\]|\\.)*)"\)', src)
    assert len(lits) == 3, len(lits)
    expected = re.findall(r'command != "([^"]*)"', src)
    stored = re.findall(r'if q != "([^"]*)"', src)
    assert len(expected) == 3 and len(stored) == 1
    pid = int(re.search(r"var pid uint32 = (\d+)", src).group(1))
    fd = int(re.search(r"var fd uint64 = (\d+)", src).group(1))
    payloads = []
    for lit in lits:
        b = go_string(lit)
        p = (b + bytes(1024))[:1024]          # p := [1024]uint8{}; copy(p[:], bytes); PayloadSize: 1024
        payloads.append(p.hex())
    kat = {
        "source": "aggregator/pg_test.go @ 2024_10_08",
        "pid": pid, "fd": fd, "method": "EXTENDED_QUERY", "payload_size": 1024,
        "known_stmt": {"parse_payload": payloads[0], "parse_expected": expected[0], "stored_query": stored[0],
                       "bind_payload": payloads[1], "bind_expected": expected[1]},
        "unknown_stmt": {"bind_payload": payloads[2], "bind_expected": expected[2]},
    }
    json.dump(kat, open(os.path.join(HERE, "pg_kat.json"), "w"), indent=1)

    cfg = json.load(open(os.path.join(REF, "testconfig/config1.json")))
    sim_src = open(os.path.join(REF, "main_benchmark_test.go"), encoding="utf-8").read()
    payload = re.search(r'payload := "([^"]*)"', sim_src)
    sim = {
        "source": "main_benchmark_test.go:561-617 + testconfig/config1.json @ 2024_10_08",
        "config": cfg,
        "payload": payload.group(1) if payload else "GET /user HTTP1.1",
        "status": 200, "duration": 50, "write_time_offset_ns": 10,
    }
    json.dump(sim, open(os.path.join(HERE, "sim_kat.json"), "w"), indent=1)
    print("wrote pg_kat.json, sim_kat.json")
    hpack_fixture()


def hpack_fixture(independent_copy: str = ""):
    """hpack_huffman_rfc7541.json: RFC 7541 Appendix B as (code, length) per symbol.  The data is the RFC's, not the
    reference repository's (HPACK is golang.org/x/net there, not vendored).  Written from the oracle's length table
    (oracle/http2.c, canonical code) and — when a path to python-hpack's huffman_constants.py is given, as was done when
    the fixture was first written — refused unless all 257 entries agree with that independent copy."""
    import importlib.util, sys
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import pyoracle
    tab = [list(pyoracle.huff_code(s)) for s in range(257)]
    if independent_copy:
        spec = importlib.util.spec_from_file_location("hc", independent_copy)
        hc = importlib.util.module_from_spec(spec); spec.loader.exec_module(hc)
        assert [[c, l] for c, l in zip(hc.REQUEST_CODES, hc.REQUEST_CODES_LENGTH)] == tab
    json.dump({"source": "RFC 7541 Appendix B (symbol -> [code, bit length], 256 = EOS); written from oracle/http2.c's length table after "
                         "checking all 257 entries against an independent copy of the appendix (python-hpack's huffman_constants.py, "
                         "found in the build container's tooling; not a dependency of this repo)", "table": tab},
              open(os.path.join(HERE, "hpack_huffman_rfc7541.json"), "w"))
    print("wrote hpack_huffman_rfc7541.json")


if __name__ == "__main__":
    main()
