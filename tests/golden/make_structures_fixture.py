"""Generates tests/golden/matbench_dielectric_n20_structures.json from the reference's bundled
data file eigenn/dataset/matbench/matbench_dielectric_n20.json (the input of BASELINE.json's
config 1; format: SURVEY.md App. D): lattice, Cartesian coordinates, atomic numbers and the
target of each of the 20 structures, plus - computed here with the ORACLE's brute-force
neighbour list (oracle/eigenn_ref.py, r_cut = 5.0, full PBC as at eigenn/data/data.py:371-388)
- the number of edges, the per-atom neighbour counts and a SHA-256 of the canonical
(i, j, Sx, Sy, Sz) table.  Data only; run once in the build container:

    python tests/golden/make_structures_fixture.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SRC = "/root/reference/eigenn/dataset/matbench/matbench_dielectric_n20.json"

SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo "
           "Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb "
           "Bi Po At Rn Fr Ra Ac Th Pa U Np Pu").split()
Z = {s: i + 1 for i, s in enumerate(SYMBOLS)}


def main():
    from oracle.eigenn_ref import neighbor_list_bruteforce

    raw = json.load(open(SRC))
    out = []
    for struct, target in raw["data"]:
        cell = np.array(struct["lattice"]["matrix"], dtype=np.float64)
        pos = np.array([s["xyz"] for s in struct["sites"]], dtype=np.float64)
        z = [Z[s["species"][0]["element"]] for s in struct["sites"]]
        ei, sh = neighbor_list_bruteforce(pos, cell, True, 5.0)
        table = np.concatenate([ei.T, sh], axis=1).astype(np.int64)
        out.append(
            dict(
                cell=cell.tolist(),
                pos=pos.tolist(),
                atomic_numbers=z,
                n=float(target),
                r_cut=5.0,
                n_edges=int(ei.shape[1]),
                num_neigh=np.bincount(ei[0], minlength=len(pos)).tolist(),
                edges_sha256=hashlib.sha256(np.ascontiguousarray(table).tobytes()).hexdigest(),
            )
        )
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matbench_dielectric_n20_structures.json")
    json.dump(out, open(path, "w"))
    print("wrote", path, [len(s["pos"]) for s in out], sum(s["n_edges"] for s in out))


if __name__ == "__main__":
    main()
