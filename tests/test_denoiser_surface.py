"""Operator surface: registry names, Config fields, state-dict keys (checkpoint compatibility) and an end-to-end
DGSDenoiser.forward (DiT -> Gaussians -> batched rasterization) on the CPU emulator vs oracle DiT + oracle rasterizer."""
import numpy as np
import pytest
import torch

from dgs_amd import denoiser as dn
from dit_util import rel_l2, synth_inputs
from emu_util import emu_lib
from oracle import dit_oracle as D
from oracle import raster_oracle as RO

OBJ_CFG = dict(width=256, in_channels=9, patch_size=8, num_layers=2, ray_pe_type="relative_plk")


def test_registry_and_state_dict_keys():
    assert dn.find("diffusion-gs-model") is dn.DGSDenoiser
    assert dn.find("diffusion-gs-model-scene") is dn.DGSDenoiserScene
    for cls, scene in ((dn.DGSDenoiser, False), (dn.DGSDenoiserScene, True)):
        m = cls(OBJ_CFG, device="cpu", lib=emu_lib())
        ref = D.init_state_dict(D.Cfg(width=256, num_layers=2, scene=scene))
        sd = m.state_dict()
        assert set(sd.keys()) == set(ref.keys())
        for k in ref:
            assert tuple(sd[k].shape) == tuple(ref[k].shape), k


def test_forward_end_to_end_emulated():
    torch.manual_seed(0)
    m = dn.DGSDenoiser(OBJ_CFG, device="cpu", lib=emu_lib())
    m.reset_parameters(seed=4)
    with torch.no_grad():   # give the decoder a larger output so Gaussians differ visibly
        m.image_token_decoder.linear.weight.mul_(20.0)
    cfg = D.Cfg(width=256, num_layers=2)
    B, V, res = 1, 2, 16
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, B, V, res, seed=2)
    batch = dict(image=images, ray_o=ray_o, ray_d=ray_d, c2w=c2w, fxfycxcy=k)
    with torch.no_grad():            # the reference samples under no_grad (gaussian_diffusion.py p_sample_loop_progressive)
        rendered, gaussians = m(batch, t)
    assert rendered.shape == (B, V, 3, res, res) and len(gaussians) == B
    sd = {k_: v.detach().clone() for k_, v in m.state_dict().items()}
    ref, _ = D.image_to_gaussians(sd, cfg, images, ray_o, ray_d, t)
    for key in ("xyz", "features", "scaling", "rotation", "opacity"):
        assert rel_l2(getattr(gaussians[0], {"xyz": "_xyz", "features": "_features_dc", "scaling": "_scaling",
                                             "rotation": "_rotation", "opacity": "_opacity"}[key]).reshape(ref[key][0].shape),
                      ref[key][0]) < 3e-2, key
    # render the HIP path's own Gaussians with the oracle rasterizer: isolates the raster/camera glue from DiT rounding
    g = gaussians[0]
    view, proj, campos, tanfov = D.camera_matrices(c2w[0], k[0], res, res)
    for v in range(V):
        o = RO.RasterOracle()
        o.forward(np.ones(3, np.float32), g._xyz.numpy(), g.get_opacity.numpy(), view[v].numpy(), proj[v].numpy(),
                  campos[v].numpy(), float(tanfov[v, 0]), float(tanfov[v, 1]), res, res, shs=g._features_dc.numpy(),
                  scales=g.get_scaling.numpy(), rotations=g.get_rotation.numpy(), exp_mode=1)
        np.testing.assert_allclose(rendered[0, v].numpy(), o.get("out_color"), atol=2e-4)


def test_training_step_end_to_end_emulated():
    """loss = MSE(DGSDenoiser.forward(batch, t) renders, target) -> .backward(): DiT backward + rasterizer backward kernels
    under torch autograd must reproduce the gradients of the same loss computed with the fp32 oracle + the per-view drop-in
    binding (reference call convention)."""
    import dgs_amd.raster as R
    R._default = emu_lib() and R.RasterBackend(emu_lib())
    import diff_gaussian_rasterization as dgr
    m = dn.DGSDenoiser(OBJ_CFG, device="cpu", lib=emu_lib())
    m.reset_parameters(seed=6)
    with torch.no_grad():
        m.image_token_decoder.linear.weight.mul_(20.0)
        for n_, p_ in m.named_parameters():          # bf16-representable GEMM weights: the oracle sees the same numbers
            if p_.dim() == 2:
                p_.copy_(p_.to(torch.bfloat16).float())
    cfg = D.Cfg(width=256, num_layers=2)
    B, V, res = 1, 2, 16
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, B, V, res, seed=3)
    batch = dict(image=images, ray_o=ray_o, ray_d=ray_d, c2w=c2w, fxfycxcy=k)
    target = torch.rand(B, V, 3, res, res, generator=torch.Generator().manual_seed(1))
    rendered, _ = m(batch, t)
    loss = ((rendered - target) ** 2).mean()
    loss.backward()
    # reference: oracle DiT (autograd) + torch activations + per-view binding
    leaf = {n_: p_.detach().clone().requires_grad_(True) for n_, p_ in m.state_dict().items()}
    g, _ = D.image_to_gaussians(leaf, cfg, images, ray_o, ray_d, t)
    view, proj, campos, tanfov = D.camera_matrices(c2w, k, res, res)
    imgs = []
    for v in range(V):
        rs = dgr.GaussianRasterizationSettings(res, res, float(tanfov[0, v, 0]), float(tanfov[0, v, 1]), torch.ones(3), 1.0,
                                               view[0, v], proj[0, v], 0, campos[0, v], False, False)
        color, _ = dgr.GaussianRasterizer(rs)(g["xyz"][0], torch.zeros_like(g["xyz"][0], requires_grad=True),
                                              torch.sigmoid(g["opacity"][0]), shs=g["features"][0], scales=torch.exp(g["scaling"][0]),
                                              rotations=torch.nn.functional.normalize(g["rotation"][0]))
        imgs.append(color)
    ref_loss = ((torch.stack(imgs)[None] - target) ** 2).mean()
    ref_loss.backward()
    print('LOSS', float(loss.detach()), float(ref_loss.detach()))
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 2e-2 * abs(float(ref_loss.detach())) + 1e-6
    bad = []
    for n_, p_ in m.named_parameters():
        r = leaf[n_].grad
        e = rel_l2(p_.grad, r)
        if e > 0.1:
            bad.append((n_, e))
    assert not bad, bad[:5]


def _tiny_model(seed=6, layers=1):
    m = dn.DGSDenoiser(dict(OBJ_CFG, num_layers=layers), device="cpu", lib=emu_lib())
    m.reset_parameters(seed=seed)
    with torch.no_grad():
        m.image_token_decoder.linear.weight.mul_(20.0)
    return m


def test_img_aligned_xyz_is_differentiable():
    """The reference's loss_xyz / loss_pointsdist are built from the SECOND output of image_to_gaussians
    (systems/diffusion_gs_system.py:90-104; lambda_pointsdist is the only active term of the first 150 steps): its gradient
    must reach the parameters, and equal the gradient of the same function of xyz[:, n_gaussians:] (of which it is a view)."""
    m = _tiny_model()
    cfg = D.Cfg(width=256, num_layers=1)
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 1, 2, 16, seed=3)
    w = torch.randn(1, 2, 3, 16, 16, generator=torch.Generator().manual_seed(0))
    params, aligned = m.image_to_gaussians(images, ray_o, ray_d, t)
    assert aligned.requires_grad
    (aligned * w).sum().backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    assert float(g1["image_token_decoder.linear.weight"].abs().max()) > 0
    m.zero_grad()
    params, _ = m.image_to_gaussians(images, ray_o, ray_d, t)
    ps = m.cfg.patch_size
    pts = params.xyz[:, m.cfg.n_gaussians:].reshape(1, 2, 16 // ps, 16 // ps, ps, ps, 3).permute(0, 1, 6, 2, 4, 3, 5).reshape(1, 2, 3, 16, 16)
    assert torch.equal(pts.detach(), aligned.detach())          # the rearrange of denoiser.py:371-379
    (pts * w).sum().backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, g1[n], rtol=1e-5, atol=1e-7), n


def test_gradient_accumulation_and_pending_forwards():
    """Two backward passes before zero_grad accumulate g1 + g2 in .grad (autograd receives copies of the flat buffer's slices);
    SEVERAL training forwards may be pending -- each holds its own activation arena until its backward ran or its graph was
    dropped -- so the sum of two forwards' losses differentiates like the two losses one by one (eval mode with grad enabled too:
    frozen-norm fine-tuning, guidance); arenas that only existed for the overlap are freed again."""
    m = _tiny_model(seed=7)
    cfg = D.Cfg(width=256, num_layers=1)
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 2, 2, 16, seed=5)
    f = lambda sl: m.image_to_gaussians(images[sl], ray_o[sl], ray_d[sl], t[sl])[0]
    loss = lambda p: (p.xyz ** 2).mean() + (p.opacity ** 2).mean() + (p.features ** 2).mean()
    loss(f(slice(0, 1))).backward()
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    loss(f(slice(1, 2))).backward()
    g2 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    loss(f(slice(0, 1))).backward()
    loss(f(slice(1, 2))).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, g1[n] + g2[n], rtol=1e-5, atol=1e-8), n
    eng = m.engine()
    for mode in (m.train, m.eval):
        mode()
        m.zero_grad()
        a, b = f(slice(0, 1)), f(slice(1, 2))                   # two forwards pending at once
        assert eng.pending_backward and sum(ar["busy"] for ar in eng._train["arenas"]) == 2
        assert a.xyz.grad_fn is not None and b.xyz.grad_fn is not None
        (loss(a) + loss(b)).backward()                           # ONE backward through both
        for n, p in m.named_parameters():
            assert torch.allclose(p.grad, g1[n] + g2[n], rtol=1e-5, atol=1e-8), (n, m.training)
        assert not eng.pending_backward and len(eng._train["arenas"]) == 1      # the second arena was only needed for the overlap
    a = f(slice(0, 1))
    assert eng.pending_backward
    del a                                                       # graph dropped without backward: arena released
    import gc; gc.collect()
    assert not eng.pending_backward
    with torch.no_grad():
        assert f(slice(0, 1)).xyz.grad_fn is None


def test_differentiable_batches_above_four_run_as_chunks():
    """dgs_dit_forward_train takes <= 4 samples per call; a larger batch (the reference's scene configurations train with 12 / 24
    per rank) goes through image_to_gaussians as chunks under autograd: same outputs and the same parameter gradients as the
    samples one by one; under a trainer with in-place gradients the split is the trainer's job."""
    m = _tiny_model(seed=11)
    cfg = D.Cfg(width=256, num_layers=1)
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 6, 2, 16, seed=5)
    loss = lambda p: (p.xyz ** 2).sum() + (p.opacity ** 2).sum() + (p.features ** 2).sum()
    ref_out, ref = [], None
    for b in range(6):
        sl = slice(b, b + 1)
        p, _ = m.image_to_gaussians(images[sl], ray_o[sl], ray_d[sl], t[sl])
        ref_out.append(p.xyz.detach())
        loss(p).backward()
    ref = {n: q.grad.clone() for n, q in m.named_parameters()}
    m.zero_grad()
    p, aligned = m.image_to_gaussians(images, ray_o, ray_d, t)     # 6 samples: chunks of 4 + 2
    assert p.xyz.shape[0] == 6 and aligned.shape[0] == 6 and torch.equal(p.xyz.detach(), torch.cat(ref_out))
    loss(p).backward()
    for n, q in m.named_parameters():
        assert torch.allclose(q.grad, ref[n], rtol=2e-4, atol=1e-6 * float(ref[n].abs().max())), n
    m._grads_in_place = True
    with pytest.raises(RuntimeError, match="micro-batches"):
        m.image_to_gaussians(images, ray_o, ray_d, t)
    m._grads_in_place = False


def test_engine_survives_optimizer_steps_in_place():
    """optimizer.step() changes parameter versions: the engine must pick the new weights up IN PLACE (same engine, same
    device buffers, same flat gradient buffer) -- and an update that bypasses version counters needs the explicit refresh."""
    m = _tiny_model(seed=9)
    cfg = D.Cfg(width=256, num_layers=1)
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 1, 2, 16, seed=5)
    eng = m.engine()
    ptr = eng._keep["0.qkv_w"].data_ptr()
    opt = torch.optim.SGD(m.parameters(), lr=0.5)
    p0, _ = m.image_to_gaussians(images, ray_o, ray_d, t)
    (p0.xyz ** 2).mean().backward()
    flat_ptr = eng._train["fg"].flat.data_ptr()
    opt.step()
    with torch.no_grad():
        p1, _ = m.image_to_gaussians(images, ray_o, ray_d, t)
    assert m.engine() is eng and eng._keep["0.qkv_w"].data_ptr() == ptr and eng._train["fg"].flat.data_ptr() == flat_ptr
    assert not torch.equal(p0.xyz.detach(), p1.xyz)
    fresh = dn.DGSDenoiser(dict(OBJ_CFG, num_layers=1), device="cpu", lib=emu_lib())
    fresh.load_state_dict(m.state_dict())
    with torch.no_grad():
        p2, _ = fresh.image_to_gaussians(images, ray_o, ray_d, t)
    assert torch.equal(p1.xyz, p2.xyz)                           # refreshed engine == engine built from the new weights
    # and the transposed copies used by the backward were refreshed too
    assert torch.equal(eng._train["tkeep"]["0.fc1"], eng._keep["0.fc1_w"].t())
    m.transformer[0].mlp.fc2.weight.data = m.transformer[0].mlp.fc2.weight.data * 2.0     # bypasses the version counter
    m.refresh_engine_weights()
    with torch.no_grad():
        p3, _ = m.image_to_gaussians(images, ray_o, ray_d, t)
    assert not torch.equal(p3.xyz, p1.xyz)


def test_run_layers_matches_oracle_blocks():
    """DGSDenoiser.run_layers(start, end) (denoiser.py:441-447): blocks [start, end) on a token tensor in the reference's order
    (gaussian tokens first) under c = t_embedder(t), vs the oracle's DiTBlock restatement; composing two ranges == one."""
    m = dn.DGSDenoiser(dict(OBJ_CFG, num_layers=3), device="cpu", lib=emu_lib())
    m.reset_parameters(seed=12)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if n_.endswith(".bias"):
                p_.copy_(torch.randn_like(p_) * 0.05)
            if p_.dim() == 2:
                p_.copy_(p_.to(torch.bfloat16).float())
    sd = {k_: v.detach().clone() for k_, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    B, V, side = 2, 2, 2                       # L = 2 + 2 * 2 * 2 = 10 tokens
    L = 2 + V * side * side
    tokens = torch.randn(B, L, 256, generator=g)
    c = D.t_embed(sd, torch.tensor([17, 801]))
    ref = tokens
    for i in range(1, 3):
        ref = D.dit_block(ref, c, sd, f"transformer.{i}.", 4)
    out = m.run_layers(1, 3)(tokens, c)                    # the reference's signature: run_layers(start, end)
    assert out.shape == tokens.shape and rel_l2(out, ref) < 2e-2, rel_l2(out, ref)
    two = m.run_layers(2, 3, views=V)(m.run_layers(1, 2, views=V)(tokens, c), c)
    assert torch.equal(two, out)
    assert torch.equal(m.run_layers(1, 1)(tokens, c), tokens)                   # empty range: identity
    # a token count that is no square grid of views (5 image tokens + 2): the workspace depends on (B, L) only
    odd = torch.randn(1, 7, 256, generator=g)
    ref7 = D.dit_block(odd, c[:1], sd, "transformer.0.", 4)
    assert rel_l2(m.run_layers(0, 1)(odd, c[:1]), ref7) < 2e-2
    with pytest.raises(RuntimeError):
        m.run_layers(0, 1, views=3)(odd, c[:1])            # 5 image tokens are not 3 views of anything


@pytest.mark.parametrize("scene", [False, True], ids=["obj", "scene"])
def test_checkpoint_files_round_trip(tmp_path, scene):
    """BASELINE configs[1]'s load path (the checkpoint itself is unreachable offline): the three file layouts the reference reads --
    the Lightning checkpoint its training writes and pipline_obj.py:66-70 loads at the SYSTEM level ({'state_dict': {'shape_model.<key>'}}),
    denoiser.py:259-267's {'model': {'denoiser.<key>'}} and a flat state dict -- written by torch.save, read back through
    `pretrained_model_name_or_path` / load_state_dict(strict=True), for the obj ([2, W]) and scene ([1, 2, W]) embedding shapes; the
    engine built from the loaded weights computes what the saved model computed."""
    cls = dn.DGSDenoiserScene if scene else dn.DGSDenoiser
    cfgd = dict(OBJ_CFG, num_layers=1, ray_pe_type="plk" if scene else "relative_plk")
    src = cls(cfgd, device="cpu", lib=emu_lib())
    src.reset_parameters(seed=31)
    assert tuple(src.gaussians_pos_embedding.shape) == ((1, 2, 256) if scene else (2, 256))
    cfg = D.Cfg(width=256, num_layers=1, scene=scene, ray_pe_type=cfgd["ray_pe_type"])
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 1, 2, 16, seed=3)
    with torch.no_grad():
        want, _ = src.image_to_gaussians(images, ray_o, ray_d, t)

    class System(torch.nn.Module):                      # the reference's LightningModule keeps the denoiser as `shape_model` (:43)
        def __init__(self, m):
            super().__init__()
            self.shape_model = m
            self.loss_computer = torch.nn.Linear(2, 2)  # stands for the LPIPS weights a real checkpoint also holds

    sd = src.state_dict()
    files = {"lightning": {"state_dict": System(src).state_dict(), "epoch": 3, "global_step": 1234},
             "model": {"model": {**{"denoiser." + k_: v for k_, v in sd.items()}, "denoiser.loss_computer.w": torch.zeros(1), "other.x": torch.zeros(1)}},
             "flat": dict(sd)}
    for name, content in files.items():
        path = str(tmp_path / f"{name}.ckpt")
        torch.save(content, path)
        m = cls(dict(cfgd, pretrained_model_name_or_path=path), device="cpu", lib=emu_lib())     # denoiser.py:256-282
        for (n_, a), (_, b) in zip(m.state_dict().items(), sd.items()):
            assert torch.equal(a, b), (name, n_)
        with torch.no_grad():
            got, _ = m.image_to_gaussians(images, ray_o, ray_d, t)
        assert all(torch.equal(got[f], want[f]) for f in ("xyz", "features", "scaling", "rotation", "opacity")), name
    # pipline_obj.py:66-70: the system-level strict load of the Lightning file
    fresh = System(cls(cfgd, device="cpu", lib=emu_lib()))
    fresh.load_state_dict(torch.load(str(tmp_path / "lightning.ckpt"), map_location="cpu")["state_dict"])
    assert torch.equal(fresh.shape_model.gaussians_pos_embedding, src.gaussians_pos_embedding)
    # the other model family's checkpoint does not load (embedding shapes differ), nor does a file with a missing key
    other = (dn.DGSDenoiser if scene else dn.DGSDenoiserScene)(dict(cfgd, ray_pe_type="relative_plk" if scene else "plk"), device="cpu", lib=emu_lib())
    with pytest.raises(RuntimeError):
        other._load_pretrained(str(tmp_path / "lightning.ckpt"))
    broken = {k_: v for k_, v in sd.items() if k_ != "transformer.0.mlp.fc2.bias"}
    with pytest.raises(RuntimeError):
        cls(cfgd, device="cpu", lib=emu_lib())._load_pretrained(broken)
