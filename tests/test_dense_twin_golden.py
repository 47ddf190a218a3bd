"""The reference's own statement of what each sparse layer computes: fixtures produced by running the DENSE-mode layers its
factories pair with every scn layer (ndsis/modules/module_factory.py, `sparse=False` branches; tests/golden/
make_dense_twin_golden.py) once on a densified seeded scene.  CPU: the oracle against them.  `-m gpu`: the HIP path, single
layers through the scn surface, against them.  Forward and the gradients of input and parameters, 1e-5 / 1e-4 of the scale
(torch's dense kernels sum in another order).

Pinned by this: the arithmetic of A5-A9 as the reference itself states it.  NOT pinned: SparseConvNet's conventions (offset
enumeration / weight layout of `scn.*Convolution.weight`, BatchNorm momentum as retain fraction, rule order) -- the mappings
from torch's layouts below are this repository's; SparseConvNet is absent from the reference checkout."""
import os

import numpy as np
import pytest
import torch

from oracle import scn_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FWD_TOL, GRAD_TOL = 1e-5, 1e-4


def _load(name):
    g = np.load(os.path.join(GOLD, f"dense_twin_{name}.npz"))
    return {k: g[k] for k in g.files}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _rel(got, ref):
    ref = ref.detach().double().cpu()
    return float((got.detach().double().cpu() - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


def conv_weight_to_scn(w):
    """torch Conv3d weight [Cout, Cin, a, b, c] -> scn layout [(a*K+b)*K+c, Cin, Cout]."""
    co, ci = w.shape[:2]
    return w.permute(2, 3, 4, 1, 0).reshape(-1, ci, co).contiguous()


def deconv_weight_to_scn(w):
    """torch ConvTranspose3d weight [Cin, Cout, a, b, c] -> scn Deconvolution layout [(a*2+b)*2+c, Cin, Cout]."""
    ci, co = w.shape[:2]
    return w.permute(2, 3, 4, 0, 1).reshape(-1, ci, co).contiguous()


def _order(fix_coords, my_coords):
    """Row permutation p with my_coords[p] == fix_coords (both hold the same site set)."""
    key = lambda c: ((c[:, 3] * 4096 + c[:, 0]) * 4096 + c[:, 1]) * 4096 + c[:, 2]
    kf, km = key(np.asarray(fix_coords, np.int64)), key(np.asarray(my_coords, np.int64))
    assert len(kf) == len(km) and np.array_equal(np.sort(kf), np.sort(km))
    om = np.argsort(km)
    return om[np.searchsorted(km[om], kf)]


CONV_CASES = [("same_conv3", 3, "subm"), ("channel_changer1", 1, "subm"), ("network_in_network", 1, "subm"),
              ("downsampler2", 2, "down"), ("upsampler2", 2, "up")]


def _oracle_rules(g, k, kind):
    if kind == "subm":
        coords = g["coords_in"]
        return O.subm_rulebook(coords, k)[1], coords, coords
    if kind == "down":
        rb = O.strided_rulebook(g["coords_in"], 2)
        return rb["rules"], g["coords_in"], rb["coords"]
    rb = O.strided_rulebook(g["coords_out"], 2)              # the encoder level the deconvolution returns to
    return O.swap_rules(rb["rules"]), rb["coords"], g["coords_out"]


@pytest.mark.parametrize("name,k,kind", CONV_CASES)
def test_oracle_conv_layers_equal_the_references_dense_twins(name, k, kind):
    g = _load(name)
    rules, cin_coords, cout_coords = _oracle_rules(g, k, kind)
    pin, pout = _order(cin_coords, g["coords_in"]), _order(g["coords_out"], cout_coords)
    # oracle rows of the input level are `cin_coords`; the fixture's X is given on g["coords_in"]
    X = _t(g["X"])[_t(pin)].clone().requires_grad_()
    W = (deconv_weight_to_scn if kind == "up" else conv_weight_to_scn)(_t(g["param_weight"])).requires_grad_()
    b = _t(g["param_bias"]).clone().requires_grad_()
    Y = O.conv(X, W, b, rules, len(cout_coords))[_t(pout)]
    assert _rel(Y, _t(g["Y"])) <= FWD_TOL
    dX, dW, db = torch.autograd.grad(Y, (X, W, b), _t(g["dY"]))
    inv = np.argsort(pin)
    assert _rel(dX[_t(inv)], _t(g["dX"])) <= GRAD_TOL
    gw = (deconv_weight_to_scn if kind == "up" else conv_weight_to_scn)(_t(g["grad_weight"]))
    assert _rel(dW, gw) <= GRAD_TOL and _rel(db, _t(g["grad_bias"])) <= GRAD_TOL


@pytest.mark.parametrize("name", ["batchnorm_leaky0", "batchnorm_leaky0p2"])
def test_oracle_batchnorm_equals_the_references_dense_twin_on_a_fully_active_grid(name):
    g = _load(name)
    X = _t(g["X"]).clone().requires_grad_()
    gamma, beta = _t(g["param_0.weight"]).clone().requires_grad_(), _t(g["param_0.bias"]).clone().requires_grad_()
    rm, rv = torch.zeros(X.shape[1]), torch.ones(X.shape[1])
    Y = O.batchnorm_relu_fwd(X, gamma, beta, rm, rv, eps=float(g["eps"]), momentum=0.9, leak=float(g["leakiness"]))
    assert _rel(Y, _t(g["Y"])) <= FWD_TOL
    for a, r in zip(torch.autograd.grad(Y, (X, gamma, beta), _t(g["dY"])), (g["dX"], g["grad_0.weight"], g["grad_0.bias"])):
        assert _rel(a, _t(r)) <= GRAD_TOL


# ---------------------------------------------------------------------------------------------------------------- -m gpu
@pytest.mark.gpu
@pytest.mark.parametrize("name,k,kind", CONV_CASES)
def test_hip_conv_layers_equal_the_references_dense_twins(name, k, kind):
    import sparse_rcnn_amd as scn
    gpu = torch.device("cuda", 0)
    g = _load(name)
    batch = int(g["batch"])
    w_t, b_t = _t(g["param_weight"]), _t(g["param_bias"])
    cin, cout = (w_t.shape[0], w_t.shape[1]) if kind == "up" else (w_t.shape[1], w_t.shape[0])
    if kind == "up":
        # Deconvolution returns to the level its Convolution came from: build that level from the FINE sites first
        fine = scn.InputLayer(3, torch.tensor([int(v) for v in g["grid_out"]]), mode=0)(
            (_t(g["coords_out"]), torch.zeros(len(g["coords_out"]), 1, device=gpu), batch))
        down = scn.Convolution(3, 1, 1, 2, 2, False).to(gpu)
        coarse = down(fine)
        my_in = coarse.get_spatial_locations().numpy()
        pin = _order(my_in, g["coords_in"])
        X = _t(g["X"])[_t(pin)].to(gpu).requires_grad_()
        x = scn.SparseConvNetTensor(features=X, metadata=coarse.metadata, spatial_size=coarse.spatial_size)
        layer = scn.Deconvolution(3, cin, cout, 2, 2, True).to(gpu)
        W = deconv_weight_to_scn(w_t)
    else:
        X0 = _t(g["X"]).to(gpu).requires_grad_()
        x = scn.InputLayer(3, torch.tensor([int(v) for v in g["grid_in"]]), mode=0)((_t(g["coords_in"]), X0, batch))
        my_in = x.get_spatial_locations().numpy()
        assert np.array_equal(my_in, g["coords_in"])                      # unique sites: first-occurrence order = input order
        X, pin = X0, np.arange(len(my_in))
        if name == "network_in_network":
            layer = scn.NetworkInNetwork(cin, cout, True).to(gpu)
        elif kind == "subm":
            layer = scn.SubmanifoldConvolution(3, cin, cout, k, True).to(gpu)
        else:
            layer = scn.Convolution(3, cin, cout, 2, 2, True).to(gpu)
        W = conv_weight_to_scn(w_t)
    with torch.no_grad():
        layer.weight.copy_(W.reshape(layer.weight.shape).to(gpu))
        layer.bias.copy_(b_t.to(gpu))
    y = layer(x)
    pout = _order(g["coords_out"], y.get_spatial_locations().numpy())
    Y = y.features[_t(pout).to(gpu)]
    assert _rel(Y, _t(g["Y"])) <= FWD_TOL
    Y.backward(_t(g["dY"]).to(gpu))
    inv = np.argsort(pin)
    assert _rel(X.grad[_t(inv).to(gpu)], _t(g["dX"])) <= GRAD_TOL
    gw = (deconv_weight_to_scn if kind == "up" else conv_weight_to_scn)(_t(g["grad_weight"]))
    assert _rel(layer.weight.grad.reshape(gw.shape), gw) <= GRAD_TOL
    assert _rel(layer.bias.grad, _t(g["grad_bias"])) <= GRAD_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["batchnorm_leaky0", "batchnorm_leaky0p2"])
def test_hip_batchnorm_equals_the_references_dense_twin_on_a_fully_active_grid(name):
    import sparse_rcnn_amd as scn
    gpu = torch.device("cuda", 0)
    g = _load(name)
    X = _t(g["X"]).to(gpu).requires_grad_()
    x = scn.InputLayer(3, torch.tensor([int(v) for v in g["grid_in"]]), mode=0)((_t(g["coords_in"]), X, int(g["batch"])))
    assert np.array_equal(x.get_spatial_locations().numpy(), g["coords_in"])
    leak = float(g["leakiness"])
    layer = (scn.BatchNormLeakyReLU(X.shape[1], float(g["eps"]), 0.9, leak) if leak else
             scn.BatchNormReLU(X.shape[1], float(g["eps"]), 0.9)).to(gpu)
    with torch.no_grad():
        layer.weight.copy_(_t(g["param_0.weight"]).to(gpu))
        layer.bias.copy_(_t(g["param_0.bias"]).to(gpu))
    layer.train()
    y = layer(x)
    assert _rel(y.features, _t(g["Y"])) <= FWD_TOL
    y.features.backward(_t(g["dY"]).to(gpu))
    assert _rel(X.grad, _t(g["dX"])) <= GRAD_TOL
    assert _rel(layer.weight.grad, _t(g["grad_0.weight"])) <= GRAD_TOL and _rel(layer.bias.grad, _t(g["grad_0.bias"])) <= GRAD_TOL
