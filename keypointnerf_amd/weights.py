"""Hot-path parameters: from the reference's (unchanged) state-dict / module tree to kernel operands.

The checkpoint format is untouched: parameters are read by the reference's own
names — ``mlp_geo.layers1.layers.N.linear.{weight_g,weight_v,bias}``, ``ibr_compress_gfeat.*``,
``mlp_tex.*`` — optionally behind the Lightning prefix ``model.`` (reference src/model.py:42).

Two products:
  * ``effective_weights``: weight-norm folded, plain row-major (out,in) fp32 matrices
    (``w = g * v / ||v||_row``, reference src/utils.py:542-543 via torch.nn.utils.weight_norm dim=0);
  * ``pack_for_mfma``: the same matrices re-ordered into the A-operand stream of
    ``v_mfma_f32_32x32x2_f32`` as consumed by csrc/field_kernels.hip.
"""
import numpy as np
import torch

from .synthetic import HOTPATH_LAYERS


def _strip_prefix(sd):
    if any(k.startswith("model.") for k in sd):
        return {k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}
    return sd


def effective_weights(state_dict):
    """dict name -> (W (out,in) float32 numpy, b (out,) float32 numpy); plus 'ani_al' -> float."""
    sd = _strip_prefix(state_dict)
    out = {}
    for name, prefix, shape, wn in HOTPATH_LAYERS:
        if wn and (prefix + ".weight_g") in sd:
            g = sd[prefix + ".weight_g"].detach().float().cpu()
            v = sd[prefix + ".weight_v"].detach().float().cpu()
            w = torch._weight_norm(v, g, 0)
        elif (prefix + ".parametrizations.weight.original0") in sd:  # new-style weight_norm
            g = sd[prefix + ".parametrizations.weight.original0"].detach().float().cpu()
            v = sd[prefix + ".parametrizations.weight.original1"].detach().float().cpu()
            w = torch._weight_norm(v, g, 0)
        else:
            w = sd[prefix + ".weight"].detach().float().cpu()
        b = sd[prefix + ".bias"].detach().float().cpu()
        if tuple(w.shape) != tuple(shape) or b.shape[0] != shape[0]:
            raise ValueError(
                f"unsupported architecture: {prefix} has shape {tuple(w.shape)}, the HIP path is built "
                f"for {shape} (configs/zju.json of the reference)")
        out[name] = (np.ascontiguousarray(w.numpy(), dtype=np.float32),
                     np.ascontiguousarray(b.numpy(), dtype=np.float32))
    out["ani_al"] = float(sd["mlp_tex.ani_al"].detach().float().cpu())
    return out


def flatten_plain(eff):
    """Flat fp32 vector in HOTPATH_LAYERS order: for each layer W row-major then b; then |ani_al|
    is NOT folded: the raw ani_al is appended last.  This is the layout the C oracle reads
    (oracle/kpnerf_oracle.c: struct kpo_weights offsets are derived from the same table)."""
    parts = []
    for name, _, _, _ in HOTPATH_LAYERS:
        w, b = eff[name]
        parts.append(w.reshape(-1))
        parts.append(b.reshape(-1))
    parts.append(np.array([eff["ani_al"]], dtype=np.float32))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def plain_grads_to_state_dict(state_dict, d_plain):
    """Gradient w.r.t. the flat effective parameters (the layout of ``flatten_plain``; what
    ``kpn_geo_rows_backward`` accumulates) -> gradient w.r.t. the reference's own parameters
    (``weight_g`` / ``weight_v`` / ``bias`` / ``weight``), by differentiating the weight-norm fold
    ``torch._weight_norm(v, g, 0)`` (reference src/utils.py:542-543).  Returns {state-dict key: grad tensor}."""
    sd = _strip_prefix(state_dict)
    d_plain = torch.as_tensor(d_plain).detach().float().cpu()
    grads, off = {}, 0
    for name, prefix, shape, wn in HOTPATH_LAYERS:
        n = shape[0] * shape[1]
        dW = d_plain[off:off + n].reshape(shape); off += n
        db = d_plain[off:off + shape[0]]; off += shape[0]
        grads[prefix + ".bias"] = db.clone()
        if wn and (prefix + ".weight_g") in sd:
            g = sd[prefix + ".weight_g"].detach().float().cpu().requires_grad_(True)
            v = sd[prefix + ".weight_v"].detach().float().cpu().requires_grad_(True)
            dg, dv = torch.autograd.grad(torch._weight_norm(v, g, 0), [g, v], dW)
            grads[prefix + ".weight_g"], grads[prefix + ".weight_v"] = dg, dv
        elif (prefix + ".parametrizations.weight.original0") in sd:
            g = sd[prefix + ".parametrizations.weight.original0"].detach().float().cpu().requires_grad_(True)
            v = sd[prefix + ".parametrizations.weight.original1"].detach().float().cpu().requires_grad_(True)
            dg, dv = torch.autograd.grad(torch._weight_norm(v, g, 0), [g, v], dW)
            grads[prefix + ".parametrizations.weight.original0"], grads[prefix + ".parametrizations.weight.original1"] = dg, dv
        else:
            grads[prefix + ".weight"] = dW.clone()
    grads["mlp_tex.ani_al"] = d_plain[off:off + 1].clone()
    return grads


def plain_tensor_from_module(net):
    """The flat effective-parameter vector (``flatten_plain`` layout) as a DIFFERENTIABLE torch tensor built from a
    live module's parameters: weight-norm folded with ``torch._weight_norm`` (reference src/utils.py:542-543), raw
    ``ani_al`` last.  Gradients of a loss w.r.t. this tensor reach ``weight_g`` / ``weight_v`` / ``weight`` / ``bias``
    through autograd — this is how the training drop-in hands the library's ``d_plain`` to the optimizer."""
    params = dict(net.named_parameters())
    if any(k.startswith("model.") for k in params):
        params = {k[len("model."):]: v for k, v in params.items() if k.startswith("model.")}
    parts = []
    for name, prefix, shape, wn in HOTPATH_LAYERS:
        if (prefix + ".weight_g") in params:
            w = torch._weight_norm(params[prefix + ".weight_v"], params[prefix + ".weight_g"], 0)
        elif (prefix + ".parametrizations.weight.original0") in params:
            w = torch._weight_norm(params[prefix + ".parametrizations.weight.original1"],
                                   params[prefix + ".parametrizations.weight.original0"], 0)
        else:
            w = params[prefix + ".weight"]
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"unsupported architecture: {prefix} has shape {tuple(w.shape)}, expected {shape}")
        parts += [w.reshape(-1).float(), params[prefix + ".bias"].reshape(-1).float()]
    parts.append(params["mlp_tex.ani_al"].reshape(1).float())
    return torch.cat(parts)
