"""Inventory of what the reference's example tree imports from `sample_factory` (test infrastructure, run where
/root/reference exists):  python -m oracle.gen_import_surface  ->  tests/golden/reference_example_imports.json
= {"<module>": {"<name>": ["<example file>", ...]}} over every .py under /root/reference/sf_examples.
tests/test_plugin_surface.py holds this engine's import surface to it: every entry must resolve, except the modules the
test lists as outside the hot-path scope (SURVEY.md §8 / DESIGN.md §7), each with its reason."""
import ast
import json
import os
import sys

REF = "/root/reference/sf_examples"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_example_imports.json")


def main():
    inv = {}
    for dp, _dn, fn in sorted(os.walk(REF)):
        for f in sorted(fn):
            if not f.endswith(".py"):
                continue
            p = os.path.join(dp, f)
            try:
                tree = ast.parse(open(p).read())
            except SyntaxError:
                continue
            rel = os.path.relpath(p, REF)
            for node in ast.walk(tree):
                if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "sample_factory":
                    for a in node.names:
                        inv.setdefault(node.module, {}).setdefault(a.name, []).append(rel)
                elif isinstance(node, ast.Import):
                    for a in node.names:
                        if a.name.split(".")[0] == "sample_factory":
                            inv.setdefault(a.name, {}).setdefault("", []).append(rel)
    inv = {m: {n: sorted(set(fs)) for n, fs in sorted(v.items())} for m, v in sorted(inv.items())}
    with open(OUT, "w") as f:
        json.dump(inv, f, indent=1, sort_keys=True)
        f.write("\n")
    print(f"{OUT}: {len(inv)} modules, {sum(len(v) for v in inv.values())} names", file=sys.stderr)


if __name__ == "__main__":
    main()
