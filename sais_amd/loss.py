"""SupCon / prototype-cosine head of SAIS on MI355X — drop-in for
SAIS/scripts/prepare_miscellaneous.py:14-46 (`calcNCELoss`), :111-126 (`getProbs`) and
process_inference_results.py:76-91 (`calcProbs`).  One single-workgroup fp32 kernel (`sais_nce`)
computes the L2 norms, the cosine "class logits", softmax probabilities, the loss and — when
gradients are needed — d emb and d prototypes in the same launch."""
import torch

from . import _lib as L
from . import ops


def nce_max_rows(C):
    """Largest batch one `sais_nce` launch takes: it is a single-workgroup kernel that keeps (2B + C + 2BC) floats in
    64 KiB of LDS (sais_amd/csrc/temporal.hip).  C = 2 -> 2730 rows."""
    return (16384 - C) // (2 + 2 * C)


def _stack(prototypes):
    keys = list(prototypes.keys())
    return torch.cat([prototypes[k].reshape(1, -1) for k in keys], dim=0).float().contiguous(), keys


def label_columns(labels, gesture_prototypes, device):
    """int32 device tensor of prototype column indices for `labels` (pass it to calcNCELoss as `labels` to keep the
    call free of host-side work, e.g. inside a captured hipGraph)."""
    return _label_cols(labels, list(gesture_prototypes.keys()), device)


def _label_cols(labels, keys, device):
    # column of the prototype whose key == str(label)  (prepare_miscellaneous.py:31-37)
    cols = [keys.index(str(int(l))) for l in (labels.tolist() if torch.is_tensor(labels) else labels)]
    return torch.tensor(cols, dtype=torch.int32, device=device)


class _NCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, cols, *protos):
        p = torch.cat([q.reshape(1, -1) for q in protos], dim=0).float().contiguous()
        emb = emb.float().contiguous()
        B, C = emb.shape[0], p.shape[0]
        if B > nce_max_rows(C):
            raise L.SaisHipError(f"calcNCELoss: batch of {B} rows with {C} prototypes exceeds the single-workgroup loss "
                                 f"kernel's limit of {nce_max_rows(C)} rows (the loss is a mean over the batch and its "
                                 "gradient couples every row with the prototypes); use a smaller batch")
        loss = torch.empty(1, dtype=torch.float32, device=emb.device)
        demb = torch.empty_like(emb)
        dpro = torch.zeros_like(p)
        ops.nce(emb, p, cols, None, None, loss, demb, dpro, 1.0)
        ctx.save_for_backward(demb, dpro)
        ctx.shapes = [q.shape for q in protos]
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        demb, dpro = ctx.saved_tensors
        return (demb * g, None) + tuple((dpro[i] * g).view(s) for i, s in enumerate(ctx.shapes))


def calcNCELoss(rank, snip_sequence, labels, videoname, gesture_prototypes, domains):
    """Same signature as the reference; `rank`, `videoname`, `domains` are accepted and unused, as there."""
    if not snip_sequence.is_cuda:
        raise L.SaisHipError("calcNCELoss needs device tensors: the HIP path has no CPU fallback")
    keys = list(gesture_prototypes.keys())
    if torch.is_tensor(labels) and labels.is_cuda and labels.dtype == torch.int32:
        cols = labels          # already prototype COLUMN indices on the device (no host work: graph-capturable)
    else:
        cols = _label_cols(labels, keys, snip_sequence.device)
    return _NCEFn.apply(snip_sequence, cols, *[gesture_prototypes[k] for k in keys])


def cosine_logits_and_probs(snip_sequence, gesture_prototypes):
    """sim [B,C] (the class logits) and probs = softmax(sim) — getProbs / calcProbs."""
    p, _ = _stack(gesture_prototypes)
    emb = snip_sequence.detach().float().contiguous()
    B, C = emb.shape[0], p.shape[0]
    sim = torch.empty(B, C, dtype=torch.float32, device=emb.device)
    probs = torch.empty_like(sim)
    # the forward is row-independent: long videos (thousands of windows) go through in chunks of the kernel's row limit
    step = nce_max_rows(C)
    for i in range(0, B, step):
        ops.nce(emb[i:i + step], p.detach(), None, sim[i:i + step], probs[i:i + step], None, None, None, 1.0)
    return sim, probs


class _ImportanceLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ipad_u8, labels_i32):
        B, S = logits.shape
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        dlog = torch.empty_like(logits)
        ops.importance_loss(logits, target, ipad_u8, labels_i32, B, S - 1, loss, dlog, 1.0)
        ctx.save_for_backward(dlog)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return dlog * g, None, None, None


def calcImportanceLoss(output_importances, importances, ipad, labels):
    """Same signature as prepare_miscellaneous.calcImportanceLoss (:48-60): output_importances [B,1,T+1,1] (from
    fullModel with importance_loss=True), importances [B,1,T] targets, ipad bool [B,1,T+1] (True = masked), labels [B]."""
    if not output_importances.is_cuda:
        raise L.SaisHipError("calcImportanceLoss needs device tensors: the HIP path has no CPU fallback")
    dev = output_importances.device
    B, S = output_importances.shape[0], output_importances.shape[2]
    logits = output_importances.reshape(B, S).float().contiguous()
    target = importances.reshape(B, S - 1).to(dev, torch.float32).contiguous()
    mask = ipad.reshape(B, S).to(dev, torch.uint8).contiguous()
    lab = labels.to(dev, torch.int32).contiguous()
    return _ImportanceLossFn.apply(logits, target, mask, lab)
