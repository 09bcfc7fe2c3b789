"""One tiny fwd+bwd+SGD step of the whole hot path on cuda:0, checked against the CPU oracle.
Called by __graft_entry__.smoke().  Test infrastructure: the oracle is imported HERE as the checker only; nothing under
sais_amd/ imports oracle/."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import synth
    from oracle import sais_oracle as O
    from sais_amd.loss import calcNCELoss, cosine_logits_and_probs
    from sais_amd.optim import SGD
    from sais_amd.temporal import fullModel
    from sais_amd.vit import vit_small

    dev = torch.device("cuda:0")
    B, T, C = 2, 3, 2
    vsd, tsd = synth.vit_state_dict(seed=0), synth.temporal_state_dict(seed=1)
    vit = vit_small(patch_size=16)
    vit.load_state_dict(vsd, strict=True)
    vit = vit.to(dev).train()
    m = fullModel('reps', C, 'in_vs_out', 384, 'ViT', modalities='RGB-Flow')
    m.load_state_dict(tsd, strict=True)
    m.dropout_p = 0.0                      # the oracle comparison below is dropout-free
    m = m.to(dev).train()
    protos = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone().to(dev)) for k, v in synth.prototypes(2, C).items()})
    clips, fclips = synth.clips(seed=1, B=B, T=T), synth.clips(seed=2, B=B, T=T)
    lens = [T, T - 1]
    pad = synth.padding_mask(lens)
    lab = synth.labels(seed=3, B=B, nclasses=C)

    opt = SGD(list(vit.parameters()) + list(m.parameters()) + list(protos.values()), lr=0.1, engines=[vit, m])
    opt.zero_grad()
    frames = torch.cat([clips, fclips]).view(2 * B * T, 3, 224, 224).to(dev)
    reps = vit(frames).view(2, B, 1, T, 384)
    emb, attn = m(reps[0], reps[1], lens, lens, 'Prototypes', pad.to(dev), pad.to(dev), None)
    loss = calcNCELoss(0, emb, lab, ["a_0", "b_1"], protos, None)
    loss.backward()
    sim, probs = cosine_logits_and_probs(emb, protos)
    gnorm = vit.blocks[0].attn.qkv.weight.grad.norm().item()
    opt.step()
    torch.cuda.synchronize()

    # checker: CPU oracle on the same inputs
    with torch.no_grad():
        _, emb_ref, attn_ref = O.e2e_forward(vsd, tsd, clips, fclips, pad, "RGB-Flow")
        sim_ref = O.cosine_logits(emb_ref, synth.prototypes(2, C))
        loss_ref = O.nce_loss(emb_ref, lab, synth.prototypes(2, C))
    dsim = (sim.cpu() - sim_ref).abs().max().item()
    dattn = (attn.cpu() - attn_ref).abs().max().item()
    print(f"[smoke] loss {loss.item():.6f} (oracle {loss_ref.item():.6f})  max|dlogit| {dsim:.2e}  "
          f"max|dattn| {dattn:.2e}  |g(qkv0)| {gnorm:.3e}")
    assert dsim <= 1e-3 and dattn <= 2e-3 and abs(loss.item() - loss_ref.item()) <= 1e-3
    assert gnorm > 0 and torch.isfinite(torch.tensor(gnorm))
