#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run once, in the build container).

Inputs come from two data files of the reference tree (never from its code, which cannot be
imported here: torch_geometric is absent):
  * autoencoder/pdb_files/input_chig/chig.pdb        - chignolin, 138 ATOM records (text)
  * autoencoder/pdb_files/graphs_2/1HP3_1.pdb.pt     - the smallest bundled graph; read as raw
    tensor storages from the zip archive, nothing is unpickled
Expected outputs are produced by the CPU oracle (oracle/), i.e. they pin the oracle against
drift and give the GPU tests full-precision targets on real inputs; they are NOT outputs of
an executed PyG ("parity unpinned" in that sense).

  python tests/golden/make_fixtures.py [/root/reference]
"""
import os
import sys
import zipfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import models as OM  # noqa: E402
from oracle import pyg_ops as O  # noqa: E402
from tapes import chig_batch, custom_pool_tape, hp3_batch, pool_tape, skipcon_tape, summarize_tensor  # noqa: E402


def chig_graph(ref_root):
    """Node features as gvae/create_graphs_for_gvae.py:67-131 builds them: one-hot(atom name) +
    one-hot(residue name) + 3 force columns; consecutive-ATOM edges (:99-111).  The reference orders
    the one-hot categories by set() iteration (:36-48, hash dependent); sorted order is used here."""
    atoms, residues = [], []
    with open(os.path.join(ref_root, "autoencoder/pdb_files/input_chig/chig.pdb")) as f:
        for line in f:
            if line.startswith("ATOM"):
                atoms.append(line[12:16].strip())
                residues.append(line[17:20].strip())
    a_types, r_types = sorted(set(atoms)), sorted(set(residues))
    a_idx = np.array([a_types.index(a) for a in atoms], dtype=np.int16)
    r_idx = np.array([r_types.index(r) for r in residues], dtype=np.int16)
    return a_idx, r_idx, len(a_types), len(r_types)


def read_pt_raw(path, num_features):
    with zipfile.ZipFile(path) as zf:
        names = {os.path.basename(os.path.dirname(n)) + "/" + os.path.basename(n): n for n in zf.namelist()}
        raw = {k: zf.read(names[f"data/{k}"]) for k in ("0", "1", "2")}
    x = np.frombuffer(raw["0"], dtype="<f4").reshape(-1, num_features).copy()
    ei = np.frombuffer(raw["1"], dtype="<i8").reshape(-1, 2).T.copy()
    ea = np.frombuffer(raw["2"], dtype="<f4").reshape(-1, 1).copy()
    return x, ei, ea


def summarize(model, loss_tuple, extra):
    out = {"loss": np.float64(loss_tuple[0].item()), "recon": np.float64(loss_tuple[1].item()),
           "kl": np.float64(loss_tuple[2].item())}
    for name, p in model.named_parameters():
        if p.grad is not None:
            out["grad/" + name] = p.grad.numpy().copy() if p.numel() <= 4096 else summarize_tensor(p.grad)
    for name, b in model.named_buffers():
        out["buf/" + name] = b.numpy().copy()
    out.update(extra)
    return out


def main():
    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    a_idx, r_idx, n_a, n_r = chig_graph(ref_root)
    assert len(a_idx) == 138 and n_a == 50 and n_r == 7, (len(a_idx), n_a, n_r)
    np.savez_compressed(os.path.join(HERE, "chig_atoms.npz"), atom_index=a_idx, residue_index=r_idx,
                        n_atom_types=n_a, n_residue_types=n_r)
    x, ei, ea = read_pt_raw(os.path.join(ref_root, "autoencoder/pdb_files/graphs_2/1HP3_1.pdb.pt"), 6)
    assert x.shape == (454, 6) and ei.shape == (2, 453)
    assert (ei[0] == np.arange(453)).all() and (ei[1] == np.arange(1, 454)).all()  # a directed path
    np.savez_compressed(os.path.join(HERE, "1hp3_1.npz"), x=x, edge_index=ei, edge_attr=ea)

    # --- skipcon and attention on 4 chignolin graphs (552 nodes, 548 edges, F_in = 60)
    b = chig_batch(4)
    n = b.x.size(0)
    for tag, cls, h, p, gat in (("skipcon", OM.SkipconVAE, 138, 0.5, None),
                                ("attention", OM.SkipconAttentionVAE, 256, 0.49, b.edge_index.size(1) + n)):
        torch.manual_seed(0)
        model = cls(60, h, 28).train()
        init_sum = float(sum(v.double().sum() for v in model.state_dict().values()))
        tape = skipcon_tape(n, h, 28, 10, p, torch.Generator().manual_seed(77), gat_edges=gat)
        lt = OM.skipcon_loss(model, b.x, b.edge_index, b.batch, 1.0, O.NoiseTape(tape))
        lt[0].backward()
        with torch.no_grad():
            model.eval()
            eps = torch.randn(10, n, 28, generator=torch.Generator().manual_seed(78))
            rec, mean, ls = model(b.x, b.edge_index, b.batch, O.NoiseTape([eps]))
        res = summarize(model, lt, {"eval/recon": rec.numpy(), "eval/mean": mean.numpy(), "eval/log_std": ls.numpy()})
        np.savez_compressed(os.path.join(HERE, f"golden_{tag}.npz"), **res, init_checksum=np.float64(init_sum))

    # --- pool variants on the real 1HP3_1 graph (all 6 columns standardised, vae_with_pool.py:253)
    pb = hp3_batch()
    n = pb.x.size(0)
    for tag, cls, lossf, tapef in (("pool", OM.PoolVAE, lambda m, t: OM.pool_loss(m, pb.x, pb.edge_index, pb.batch, 1.0, t), pool_tape),
                                   ("custom_pool", OM.CustomPoolVAE, lambda m, t: OM.custom_pool_loss(m, pb.x, pb.edge_index, pb.batch, t), custom_pool_tape)):
        torch.manual_seed(0)
        model = cls(6, 138, 28).train()
        init_sum = float(sum(v.double().sum() for v in model.state_dict().values()))
        tape = tapef(n, 138, 28, 10, torch.Generator().manual_seed(79))
        lt = lossf(model, O.NoiseTape(tape))
        lt[0].backward()
        with torch.no_grad():
            model.eval()
            eps = torch.randn(10, n, 28, generator=torch.Generator().manual_seed(80))
            out = model(pb.x, pb.edge_index, pb.batch, O.NoiseTape([eps]))
        res = summarize(model, lt, {"eval/recon": out[0].numpy(), "eval/x_pool": out[1].numpy(),
                                    "eval/edge_index_pool": out[2].numpy()})
        np.savez_compressed(os.path.join(HERE, f"golden_{tag}.npz"), **res, init_checksum=np.float64(init_sum))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
