"""PDB -> graph builder (gvae/create_graphs_for_gvae.py:67-144) against the chignolin fixture: the 138 ATOM records of the
reference's autoencoder/pdb_files/input_chig/chig.pdb as arrays, with the edges / distances the reference's builder makes of
them (tests/golden/make_pdb_fixture.py)."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture():
    return np.load(os.path.join(GOLDEN, "chig_pdb.npz"), allow_pickle=False)


def pdb_lines(d, serial_offset=0):
    """ATOM lines in the fixed PDB columns, rebuilt from the fixture's arrays (plus a header and a TER / END tail)."""
    out = ["REMARK   1 SYNTHESISED FROM ARRAYS\n"]
    for i in range(len(d["serial"])):
        name = d["atom_name"][i].decode()
        name = (" " + name).ljust(4) if len(name) < 4 else name
        x, y, z = d["coords"][i]
        out.append("ATOM  %5d %4s %3s A%4d    %8.3f%8.3f%8.3f  1.00  0.00\n"
                   % (int(d["serial"][i]) + serial_offset, name, d["residue_name"][i].decode(), 1 + i // 14, x, y, z))
    out += ["TER\n", "END\n"]
    return out


def test_parser_reads_the_fixed_columns():
    from hgvae_amd.pdb import parse_pdb, unique_types
    d = fixture()
    rec = parse_pdb(pdb_lines(d))
    assert len(rec) == 138
    assert rec.serial.tolist() == d["serial"].tolist()
    assert rec.atom_name == [s.decode() for s in d["atom_name"]]
    assert rec.residue_name == [s.decode() for s in d["residue_name"]]
    assert np.array_equal(rec.coords, d["coords"])          # three decimals survive the text round trip exactly
    a, r = unique_types([rec])
    assert a == [s.decode() for s in d["atom_types"]] and r == [s.decode() for s in d["residue_types"]]


def test_forces_file_pattern(tmp_path):
    from hgvae_amd.pdb import load_forces
    p = tmp_path / "a_forces.txt"
    p.write_text("Vec3(x=-1.5, y=2.25, z=10.0) kJ/(nm mol)\n[0.5 -0.125 3.0]\n")
    # the reference's pattern r"[-+]?\\d*\\.\\d+|\\d+" also picks the digit of "Vec3": first three matches per line
    f = load_forces(str(p))
    assert f.shape == (2, 3) and f[0].tolist() == [3.0, -1.5, 2.25] and f[1].tolist() == [0.5, -0.125, 3.0]


def test_repeated_serials_follow_the_reference_dictionary_logic():
    from hgvae_amd.pdb import _edges_with_repeated_serials
    # serials 1 2 3 2 4: the map keeps the LAST node of serial 2 (node 3), so records 1->2 and 2->3 attach to node 3
    e = _edges_with_repeated_serials(np.array([1, 2, 3, 2, 4]))
    assert e.T.tolist() == [[0, 3], [2, 3], [3, 4]]


@pytest.mark.gpu
def test_device_builder_matches_the_fixture_bit_for_bit():
    from hgvae_amd.pdb import build_shard, parse_pdb
    d = fixture()
    dev = torch.device("cuda:0")
    recs = [parse_pdb(pdb_lines(d)), parse_pdb(pdb_lines(d, serial_offset=1000)), parse_pdb(pdb_lines(d))]
    gen = np.random.default_rng(0)
    forces = [gen.normal(size=(138, 3)) * 50.0, None, gen.normal(size=(138, 3))]
    shard = build_shard(recs, dev, forces)
    n_a, n_r = len(d["atom_types"]), len(d["residue_types"])
    assert shard.num_features == n_a + n_r + 3 == 60 and len(shard) == 3
    x = shard.x.cpu()
    for g in range(3):
        xs = x[138 * g:138 * (g + 1)]
        want = torch.zeros(138, 60)
        want[torch.arange(138), torch.from_numpy(d["atom_index"]).long()] = 1
        want[torch.arange(138), n_a + torch.from_numpy(d["residue_index"]).long()] = 1
        if forces[g] is not None:
            want[:, 57:] = torch.from_numpy(forces[g].astype(np.float32))
        assert torch.equal(xs, want), g
        lo, hi = 137 * g, 137 * (g + 1)
        assert torch.equal(shard.edge_index[:, lo:hi].cpu(), torch.from_numpy(d["edge_index"]) + 138 * g)
        assert torch.equal(shard.edge_attr[lo:hi, 0].cpu(), torch.from_numpy(d["edge_attr"]))
    # a batch drawn from the shard is the host-collated batch of the same graphs, CSR included
    b = shard.batch([2, 0])
    assert b.x.shape == (276, 60) and b.edge_index.shape == (2, 274) and b.num_graphs == 2
    assert torch.equal(b.x[:138].cpu(), x[276:414]) and torch.equal(b.edge_index[:, 137:].cpu(), torch.from_numpy(d["edge_index"]) + 138)
    b.graph.check()


@pytest.mark.gpu
def test_device_builder_standardises_force_columns_and_handles_repeated_serials():
    from hgvae_amd.pdb import AtomRecords, build_shard
    dev = torch.device("cuda:0")
    gen = np.random.default_rng(1)
    a = AtomRecords([1, 2, 3, 4], ["N", "CA", "C", "O"], ["GLY"] * 4, [1] * 4, gen.normal(size=(4, 3)))
    b = AtomRecords([1, 2, 3, 2, 4], ["N", "CA", "C", "CA", "O"], ["ALA"] * 5, [1] * 5, gen.normal(size=(5, 3)))
    fa, fb = gen.normal(size=(4, 3)) * 30 + 5, gen.normal(size=(5, 3)) * 30 + 5
    shard = build_shard([a, b], dev, [fa, fb], numerical_indices=[6, 7, 8])
    assert shard.num_features == 4 + 2 + 3
    allf = torch.from_numpy(np.concatenate([fa, fb]).astype(np.float32))
    want = (allf - allf.mean(0)) / allf.std(0)
    assert torch.allclose(shard.x[:, 6:].cpu(), want, rtol=1e-5, atol=1e-6)
    ei = shard.edge_index.cpu()
    assert ei[:, :3].T.tolist() == [[0, 1], [1, 2], [2, 3]]
    assert ei[:, 3:].T.tolist() == [[4, 7], [6, 7], [7, 8]]            # file b: nodes 4..8, repeated serial 2
    d = np.linalg.norm(b.coords[0] - b.coords[3])
    assert shard.edge_attr[3, 0].item() == np.float32(d)
