#!/usr/bin/env python3
"""Generates tests/golden/chig_pdb.npz: the ATOM records of the reference's chignolin file as plain arrays plus the
graph the reference's builder makes of them (gvae/create_graphs_for_gvae.py:67-144), restated with numpy:

  inputs   serial [138] int32, atom_name / residue_name [138] fixed-width bytes, coords [138, 3] float64 (the text
           columns 6:11, 12:16, 17:20, 30:38 / 38:46 / 46:54 of every line that starts with ATOM)
  outputs  edge_index [2, 137] (consecutive ATOM records, :99-111), edge_attr [137] = float32(np.linalg.norm(a - b))
           (:24-25, :109, :137-138), atom / residue one-hot indices in SORTED category order (the reference iterates a
           set, :36-48, which is hash dependent)

Input: autoencoder/pdb_files/input_chig/chig.pdb of the reference tree (a data file; no reference code is run).
  python tests/golden/make_pdb_fixture.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    serial, names, residues, coords = [], [], [], []
    with open(os.path.join(ref_root, "autoencoder/pdb_files/input_chig/chig.pdb")) as f:
        for line in f:
            if line.startswith("ATOM"):
                serial.append(int(line[6:11].strip()))
                names.append(line[12:16].strip())
                residues.append(line[17:20].strip())
                coords.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    coords = np.array(coords, dtype=np.float64)
    n = len(serial)
    assert n == 138 and len(set(serial)) == n
    a_types, r_types = sorted(set(names)), sorted(set(residues))
    ei = np.stack([np.arange(n - 1), np.arange(1, n)]).astype(np.int64)
    ea = np.array([np.linalg.norm(coords[i] - coords[i + 1]) for i in range(n - 1)]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "chig_pdb.npz"), serial=np.array(serial, dtype=np.int32),
                        atom_name=np.array(names, dtype="S4"), residue_name=np.array(residues, dtype="S3"), coords=coords,
                        atom_index=np.array([a_types.index(a) for a in names], dtype=np.int32),
                        residue_index=np.array([r_types.index(r) for r in residues], dtype=np.int32),
                        atom_types=np.array(a_types, dtype="S4"), residue_types=np.array(r_types, dtype="S3"),
                        edge_index=ei, edge_attr=ea)
    print("wrote chig_pdb.npz:", n, "atoms,", len(a_types), "atom names,", len(r_types), "residue names")


if __name__ == "__main__":
    main()
