"""Parses the reference's type stubs (timemachine/lib/custom_ops.pyi) into tests/golden/custom_ops_api.json:
class -> base, method -> [argument names, which have defaults]; module-level functions; module constants.

Run in the build container only (needs /root/reference):    python tests/golden/generate_api_fixture.py
The fixture is data about the boundary's SHAPE (names and arity), not source text; tests/test_api_conformance.py
introspects timemachine_amd.lib.custom_ops against it.
"""
import ast
import json
import os
import sys

REF = os.environ.get("TM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def sig(fn: ast.FunctionDef):
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args]
    n_def = len(a.defaults)
    has_default = [False] * (len(names) - n_def) + [True] * n_def
    return {"args": names, "has_default": has_default, "varargs": a.vararg is not None, "kwargs": a.kwarg is not None}


def main():
    path = os.path.join(REF, "timemachine", "lib", "custom_ops.pyi")
    tree = ast.parse(open(path).read())
    api = {"classes": {}, "functions": {}, "constants": []}
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            methods = {}
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name not in ("__buffer__", "__release_buffer__"):
                    methods[item.name] = sig(item)
            api["classes"][node.name] = {"bases": [ast.unparse(b) for b in node.bases], "methods": methods}
        elif isinstance(node, ast.FunctionDef):
            api["functions"][node.name] = sig(node)
        elif isinstance(node, ast.AnnAssign):
            api["constants"].append(node.target.id)
    out = os.path.join(HERE, "custom_ops_api.json")
    with open(out, "w") as fh:
        json.dump(api, fh, indent=1, sort_keys=True)
    print(f"{len(api['classes'])} classes, {len(api['functions'])} functions -> {out}")


if __name__ == "__main__":
    sys.exit(main())
