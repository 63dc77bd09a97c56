"""Generates tests/golden/lsdi_nmr_n20_structures.json from the reference's bundled data file
eigenn/dataset/LSDI_NMR/LSDI_NMR_tensor.json (the input of BASELINE.json's config 2; format:
SURVEY.md App. D): per entry the lattice, Cartesian coordinates, atomic numbers, the NMR tensors
and the site indices they belong to.  Data only (the unpacking into one data point per tensor,
eigenn/dataset/LSDI.py:69-88, is done by the test that reads the fixture); run once in the build
container:

    python tests/golden/make_lsdi_fixture.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_structures_fixture import Z  # noqa: E402

SRC = "/root/reference/eigenn/dataset/LSDI_NMR/LSDI_NMR_tensor.json"


def main():
    raw = json.load(open(SRC))
    out = []
    for entry in raw:
        st = entry["structure"]
        out.append(dict(
            cell=st["lattice"]["matrix"],
            pos=[s["xyz"] for s in st["sites"]],
            atomic_numbers=[Z[s["species"][0]["element"]] for s in st["sites"]],
            tensors=[t["data"] for t in entry["tensor"]],
            ind=[int(i) for i in entry["ind"]],
        ))
    path = os.path.join(HERE, "lsdi_nmr_n20_structures.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, [len(e["pos"]) for e in out], [len(e["tensors"]) for e in out])


if __name__ == "__main__":
    main()
