"""Checkpoint codec + model factory — drop-in for `loadModel` (SAIS/scripts/prepare_model.py:517-570)
and the save step of `trainModel` (train.py:98-121).  File contracts:

  params.zip      torch.save(state_dict of fullModel) with EVERY key prefixed `module.` (DDP era); may carry
                  `encoder.*` tensors of the unused timm ViT-B (ignored here, never instantiated)
  prototypes.zip  torch.save(nn.ParameterDict{'0': Parameter[1,256], ...})  — a pickled module object,
                  so torch >= 2.6 needs weights_only=False
  dino_deitsmall16_pretrain.pth   plain state_dict of vit_small, 150 tensors, loaded strict
"""
import os

import torch
import torch.nn as nn

from .optim import SGD
from .temporal import fullModel
from .vit import vit_small


def strip_module_prefix(params):
    """prepare_model.py:524-527: `name.split('module.')[1]`; a key without the prefix raises (IndexError there)."""
    out = {}
    for name, t in params.items():
        parts = name.split('module.')
        if len(parts) < 2:
            raise IndexError(f"checkpoint key {name!r} lacks the 'module.' prefix (prepare_model.py:526)")
        out[parts[1]] = t
    return out


def load_params_file(path):
    params = torch.load(path, map_location='cpu', weights_only=False)
    params = strip_module_prefix(params)
    return {k: v for k, v in params.items() if not k.startswith('encoder.')}     # timm ViT-B ballast, unused


def save_params_file(model, path):
    """What train.py:108 writes as `params` (README renames to params.zip): module.-prefixed state_dict."""
    sd = {'module.' + k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.save(sd, path)


def save_prototypes_file(prototypes, path):
    pd = nn.ParameterDict()
    for k in prototypes.keys():
        pd[k] = nn.Parameter(prototypes[k].detach().cpu().clone())
    torch.save(pd, path)


def load_prototypes_file(path, device):
    protos = torch.load(path, map_location='cpu', weights_only=False)
    out = nn.ParameterDict()
    for k in protos.keys():
        out[k] = nn.Parameter(protos[k].detach().clone().to(device))
    return out


def loadModel(rank, world_size, savepath, data_type, nclasses, domain, rep_dim, encoder_type, task, fold, lr=0.001,
              modalities='RGB-Flow', freeze_encoder_params=True, self_attention=True, importance_loss=False,
              inference=False, device=None):
    """Same positional signature and return value as the reference: ({'model','prototypes'}, optimizer, device).
    The reference pins device='cpu' (:544); here the model lives on the rank's MI355X."""
    device = torch.device(device if device is not None else f"cuda:{rank}")
    model = fullModel(data_type, nclasses, domain, rep_dim, encoder_type, modalities=modalities,
                      freeze_encoder_params=freeze_encoder_params, self_attention=self_attention,
                      importance_loss=importance_loss)
    if inference:
        new_params = load_params_file(os.path.join(savepath, 'params.zip'))
        print('# of Loaded Params: %i' % len(new_params), '# of Model Params: %i' % len(dict(model.named_parameters())))
        model.load_state_dict(new_params)                       # strict, as :529
        print('Params Loaded...')
    model.to(device)
    if not inference:
        protos = nn.ParameterDict()
        for c in range(nclasses):
            protos[str(c)] = nn.Parameter(torch.rand(1, 256, device=device))     # :556-560
    else:
        protos = load_prototypes_file(os.path.join(savepath, 'prototypes.zip'), device)
        print('Prototypes Loaded!')
    params = list(model.parameters()) + list(protos.values())
    optimizer = SGD(params, lr=lr, engines=[model])
    return {'model': model, 'prototypes': protos}, optimizer, device


def load_vit(path=None, device='cuda:0', drop_path_rate=0.1):
    """extract_representations.loadModel (:181-219): vit_small(patch_size=16) + strict load of the DINO checkpoint."""
    model = vit_small(patch_size=16, drop_path_rate=drop_path_rate)
    if path is not None:
        sd = torch.load(path, map_location='cpu', weights_only=False)
        if isinstance(sd, dict) and 'student' in sd:            # surgical-DINO variant (:190-199)
            items = list(sd['student'].items())[:-8]
            sd = {'.'.join(k.split('.')[2:]): v for k, v in items}
        model.load_state_dict(sd, strict=True)
    return model.to(device).eval()
