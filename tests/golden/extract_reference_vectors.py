"""Extract the pinned regression inputs of the reference's block-WAND differential tests into a
small JSON fixture.  Run in the authoring container only (needs /root/reference); the output
tests/golden/block_wand_regressions.json is committed and is what the tests read.

Sources (test *data* only, no code):
  src/query/boolean_query/block_wand_union.rs:506-609         test_fn_reproduce_proptest
  src/query/boolean_query/block_wand_intersection.rs:426-616  ..._three_scorers_regression
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def fn_body(src, name):
    i = src.index("fn " + name)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        if src[k] == "{":
            depth += 1
        elif src[k] == "}":
            depth -= 1
            if depth == 0:
                return src[j:k + 1]
        k += 1


def parse(body):
    # posting lists = maximal runs of "(a, b)," pairs; fieldnorms = the bare integer list after
    # "let fieldnorms"
    fn_i = body.index("let fieldnorms")
    lists_src, fn_src = body[:fn_i], body[fn_i:]
    lists = []
    for chunk in re.split(r"\]\s*,", lists_src):
        pairs = re.findall(r"\((\d+),\s*(\d+)\)", chunk)
        if pairs:
            lists.append([[int(a), int(b)] for a, b in pairs])
    fn_list = re.search(r"\[([\d,\s]+)\]", fn_src).group(1)
    fieldnorms = [int(x) for x in re.findall(r"\d+", fn_list)]
    return {"posting_lists": lists, "fieldnorms": fieldnorms}


def main():
    out = {}
    u = open(os.path.join(REF, "src/query/boolean_query/block_wand_union.rs")).read()
    out["union_reproduce_proptest"] = parse(fn_body(u, "test_fn_reproduce_proptest"))
    i = open(os.path.join(REF, "src/query/boolean_query/block_wand_intersection.rs")).read()
    out["intersection_three_scorers_regression"] = parse(
        fn_body(i, "test_block_wand_intersection_three_scorers_regression"))
    for k, v in out.items():
        print(k, [len(l) for l in v["posting_lists"]], len(v["fieldnorms"]))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "block_wand_regressions.json")
    json.dump(out, open(dst, "w"))
    compat_index()


def compat_index():
    """The reference's backward-compatibility fixtures (tests/compat_tests_data/index_v6, index_v7:
    one-document indexes written by released tantivy versions): file bytes as hex, a few hundred
    bytes each.  They pin the file framing, the TermInfoStore layout and real BitPacker/vint
    posting bytes for the oracle (tests/test_oracle_kat.py)."""
    out = {}
    for ver in ("index_v6", "index_v7"):
        d = os.path.join(REF, "tests", "compat_tests_data", ver)
        meta = json.load(open(os.path.join(d, "meta.json")))
        seg = meta["segments"][0]["segment_id"].replace("-", "")
        files = {}
        for ext in ("idx", "pos", "term", "fieldnorm"):
            files[ext] = open(os.path.join(d, seg + "." + ext), "rb").read().hex()
        out[ver] = {"max_doc": meta["segments"][0]["max_doc"],
                    "schema": [{"name": f["name"], "type": f["type"],
                                "record": f["options"].get("indexing", {}).get("record")
                                if isinstance(f["options"].get("indexing"), dict) else None}
                               for f in meta["schema"]],
                    "files": files}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat_index.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("compat_index:", {k: {e: len(v) // 2 for e, v in o["files"].items()} for k, o in out.items()})


if __name__ == "__main__":
    main()
