"""Generates tests/golden/data_split_golden.json in THIS container by running the REFERENCE's own
`data_split` (/root/reference/utils.py:36-61).  The module cannot be imported (pymatgen / skimage /
func_timeout at its top are absent), so the one FunctionDef is pulled out of the file with `ast`,
compiled and executed with only `os` and `random` in scope -- it is the reference's code that runs,
nothing of its text is written to the repo.  The fixture is data: directory listings in, id lists out.

    python tests/golden/make_data_split_golden.py
"""
import ast
import json
import os
import random
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/utils.py"


def reference_data_split():
    tree = ast.parse(open(REF).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "data_split"]
    assert len(fn) == 1
    ns = {"os": os, "random": random}
    exec(compile(ast.Module(body=fn, type_ignores=[]), REF, "exec"), ns)
    return ns["data_split"]


def listing(case):
    """file names under density_matrices/ for one case"""
    names = []
    for i in case["ids"]:
        names.append(i + ".npy")
        for k in range(case["rot_on_disk"]):
            names.append("%s_rot_%d.npy" % (i, k))
    return names + case.get("extra", [])


CASES = [
    {"name": "mp_ids_all", "ids": ["mp-%d" % (7 * i + 3) for i in range(23)], "rot_on_disk": 2,
     "kw": {"n": None, "frac": 0.8, "n_rot": 2}},
    {"name": "mp_ids_capped", "ids": ["mp-%d" % (7 * i + 3) for i in range(23)], "rot_on_disk": 2,
     "kw": {"n": 10, "frac": 0.8, "n_rot": 10}},          # more rotations requested than exist on disk
    {"name": "no_rotations", "ids": ["mp-%d" % i for i in range(1, 12)], "rot_on_disk": 0,
     "kw": {"n": 7, "frac": 0.5, "n_rot": 0}},
    {"name": "no_shuffle", "ids": ["mp-%d" % i for i in range(1, 12)], "rot_on_disk": 1,
     "kw": {"n": 9, "frac": 0.75, "n_rot": 1, "shuffle": False}},
    {"name": "other_seed", "ids": ["mvc-%d" % (11 * i) for i in range(40)], "rot_on_disk": 1,
     "kw": {"n": 33, "frac": 0.9, "n_rot": 3, "seed": 5}},
    # str.strip(".npy") strips characters: stems that begin/end with '.', 'n', 'p', 'y' are mangled
    {"name": "strip_quirk", "ids": ["nacl", "pyrite", "mp-12", "zn-any", "yttria.p"], "rot_on_disk": 1,
     "extra": ["README.txt", "notes.npz"], "kw": {"n": None, "frac": 0.6, "n_rot": 2}},
    {"name": "samples_20000_nrot_10", "ids": ["mp-%d" % i for i in range(300)], "rot_on_disk": 0,
     "kw": {"n": 20000, "frac": 0.8, "n_rot": 10}},
]


def main():
    ref = reference_data_split()
    out = []
    for case in CASES:
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "density_matrices"))
            files = listing(case)
            for f in files:
                open(os.path.join(tmp, "density_matrices", f), "w").close()
            tr, va = ref(tmp, **case["kw"])
        out.append({"name": case["name"], "files": files, "kwargs": case["kw"], "train": tr, "val": va})
        print(case["name"], len(files), "files ->", len(tr), "train,", len(va), "val")
    with open(os.path.join(HERE, "data_split_golden.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
