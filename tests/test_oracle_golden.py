"""Pins the CPU oracle (oracle/clip_oracle.py) to the golden vectors the REFERENCE produced
(oracle/make_golden.py).  CPU-only; runs in `-m "not gpu"`."""
import numpy as np
import pytest
import torch

from oracle import clip_oracle as O
from open_clip_amd.configs import get_model_config
from open_clip_amd.synth import init_state_dict, synthetic_batch
from tests.golden_util import check_grad, grad_keys, load, state_from_golden

TOL = 2e-5  # fp32 CPU vs fp32 CPU; differences are summation order only


def _close(a, b, tol=TOL):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("name,siglip", [("tiny_clip.npz", False), ("tiny_siglip.npz", True), ("tiny_quickgelu.npz", False), ("tiny_hd88.npz", False)])
def test_tiny_forward_backward_matches_reference(name, siglip):
    g = load(name)
    cfg = get_model_config("hd88-test" if "hd88" in name else "tiny-test")  # hd88: head_width 88, mlp_ratio 4.3637 (ViT-g-14's shape class)
    if "quickgelu" in name:  # reference CLIP(quick_gelu=True): QuickGELU in both towers (layers.py:29-32)
        cfg["quick_gelu"] = True
    state = state_from_golden(g)
    image = torch.from_numpy(g["image"].astype(np.float32))
    text = torch.from_numpy(g["text"])
    outs, grads = O.train_forward_backward(image, text, state, cfg, siglip=siglip)
    assert _close(outs["image_features"], g["out/image_features"])
    assert _close(outs["text_features"], g["out/text_features"])
    ref_logits = g["out/logits_per_image"] - (float(state["logit_bias"]) if siglip else 0.0)
    assert _close(outs["logits_per_image"], ref_logits, 1e-4)
    assert abs(float(outs["loss"]) - float(g["out/loss"])) < 1e-4 * max(1.0, abs(float(g["out/loss"])))
    worst = 0.0
    for k in grad_keys(g):
        rel, nrel = check_grad(g, k, grads[k], 0)
        worst = max(worst, rel, nrel)
        assert rel < 2e-4 and nrel < 2e-4, (k, rel, nrel)
    print("worst grad rel-L2", worst)


def test_vitb32_b8_matches_reference():
    g = load("vitb32_b8.npz")
    cfg = get_model_config("ViT-B-32")
    state = init_state_dict(cfg, seed=0, perturb=True)
    batch = synthetic_batch(cfg, 8, seed=1234)
    # the fixture stores checksums of the regenerated weights/inputs: skip (not fail) on an RNG mismatch
    for k in ("visual.conv1.weight", "token_embedding.weight", "visual.proj"):
        if abs(float(state[k].double().sum()) - float(g["wsum/" + k])) > 1e-6 * state[k].numel() ** 0.5:
            pytest.skip("torch CPU RNG stream differs from the one that generated the fixture")
    assert np.array_equal(batch["text"].numpy(), g["text"])
    outs, grads = O.train_forward_backward(batch["image"], batch["text"], state, cfg)
    assert _close(outs["image_features"], g["out/image_features"], 5e-5)
    assert _close(outs["text_features"], g["out/text_features"], 5e-5)
    assert abs(float(outs["loss"]) - float(g["out/loss"])) < 1e-4
    for k in grad_keys(g):
        rel, nrel = check_grad(g, k, grads[k], 0)
        assert rel < 1e-3 and nrel < 1e-3, (k, rel, nrel)


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_loss_semantics_match_reference(world):
    """oracle.clip_loss / siglip_loss with hand-built 'gathered' tensors reproduce what the reference's
    ClipLoss / SigLipLoss computed per rank under gloo (loss.py:29-54, :91-141, :406-489)."""
    g = load(f"dist_loss_w{world}.npz")
    feats = torch.from_numpy(g["feats"])  # [W, 2, B, E]
    for rank in range(world):
        for mode, local, gwg in (("global", False, False), ("local_gwg", True, True),
                                 ("local_nograd", True, False), ("global_gwg", False, True)):
            img = feats[rank, 0].clone().requires_grad_(True)
            txt = feats[rank, 1].clone().requires_grad_(True)
            s = torch.tensor(float(g["scale"]), requires_grad=True)
            all_img = [feats[r, 0] for r in range(world)]
            all_txt = [feats[r, 1] for r in range(world)]
            if gwg or not local:
                all_img[rank], all_txt[rank] = img, txt  # loss.py:47-50 / differentiable gather
            loss = O.clip_loss(img, txt, s, torch.cat(all_img), torch.cat(all_txt), local_loss=local, rank=rank)
            loss.backward()
            assert abs(float(loss) - float(g[f"r{rank}/clip/{mode}/loss"])) < 1e-5
            if not gwg:  # with gather_with_grad the reference adds the other ranks' contributions
                assert _close(img.grad, g[f"r{rank}/clip/{mode}/dimg"], 1e-5)
                assert _close(txt.grad, g[f"r{rank}/clip/{mode}/dtxt"], 1e-5)
                assert _close(s.grad, g[f"r{rank}/clip/{mode}/dscale"], 1e-5)
        img = feats[rank, 0].clone().requires_grad_(True)
        s = torch.tensor(float(g["scale"]), requires_grad=True)
        b = torch.tensor(float(g["bias"]), requires_grad=True)
        loss = O.siglip_loss(img, [feats[r, 1] for r in range(world)], s, b, rank)
        loss.backward()
        assert abs(float(loss) - float(g[f"r{rank}/siglip/bidir/loss"])) < 1e-5
        assert _close(img.grad, g[f"r{rank}/siglip/bidir/dimg"], 1e-5)
        assert _close(s.grad, g[f"r{rank}/siglip/bidir/dscale"], 1e-5)
        assert _close(b.grad, g[f"r{rank}/siglip/bidir/dbias"], 1e-5)


def test_pruned_towers_are_exact_in_the_oracle():
    """The two structural savings of the native towers, stated on the CPU oracle in float64 (independent of any kernel):
    (1) packed text: dropping every position behind the pooled EOT token changes nothing -- under the causal mask (transformer.py:1716-1722)
        position l only reads positions <= l, and ``text_global_pool`` (:941-944) reads x[b, argmax];
    (2) pooled last block: the poolers read one row per sequence of the last block's output, and a residual block is row-wise except
        inside the attention, so everything behind the last block's attention can run on the pooled rows alone.
    Features AND every parameter gradient of the pruned evaluation equal the reference evaluation's to float64 round-off."""
    cfg = get_model_config("tiny-test")
    state = {k: v.double() for k, v in init_state_dict(cfg, seed=4, perturb=True).items()}
    batch = synthetic_batch(cfg, 5, seed=3)
    image, text = batch["image"].double(), batch["text"]
    v, t = cfg["vision_cfg"], cfg["text_cfg"]

    def last_block_on_rows(x, p, pre, heads, causal, pick):
        """resblock (oracle/clip_oracle.py::resblock) with everything behind the attention evaluated on the rows ``pick(x)`` selects"""
        a = O.attention(O.layer_norm(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"]), p, pre, heads, causal)
        xm = pick(x) + pick(a)
        h = O.layer_norm(xm, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])
        h = O.gelu_erf(h @ p[pre + "mlp.c_fc.weight"].t() + p[pre + "mlp.c_fc.bias"])
        return xm + h @ p[pre + "mlp.c_proj.weight"].t() + p[pre + "mlp.c_proj.bias"]

    def pruned_image(p):
        ps, width = v["patch_size"], v["width"]
        B, Cin, H, W = image.shape
        gh, gw = H // ps, W // ps
        w = p["visual.conv1.weight"].reshape(width, Cin * ps * ps)
        patches = image.reshape(B, Cin, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, Cin * ps * ps)
        x = torch.cat([p["visual.class_embedding"].reshape(1, 1, width).expand(B, 1, width), patches @ w.t()], dim=1) + p["visual.positional_embedding"]
        x = O.layer_norm(x, p["visual.ln_pre.weight"], p["visual.ln_pre.bias"])
        heads = width // v.get("head_width", 64)
        x = O.transformer(x, p, "visual.transformer.", v["layers"] - 1, heads, causal=False)
        y = last_block_on_rows(x, p, f"visual.transformer.resblocks.{v['layers'] - 1}.", heads, False, lambda z: z[:, 0])
        return O.l2_normalize(O.layer_norm(y, p["visual.ln_post.weight"], p["visual.ln_post.bias"]) @ p["visual.proj"])

    def pruned_text(p):
        feats = []
        for b in range(text.shape[0]):  # one sequence at a time, cut behind its EOT token: the packed rows of the native tower
            n = int(text[b].argmax()) + 1
            x = (p["token_embedding.weight"][text[b, :n]] + p["positional_embedding"][:n]).unsqueeze(0)
            x = O.transformer(x, p, "transformer.", t["layers"] - 1, t["heads"], causal=True)
            y = last_block_on_rows(x, p, f"transformer.resblocks.{t['layers'] - 1}.", t["heads"], True, lambda z: z[:, -1])
            feats.append(O.layer_norm(y, p["ln_final.weight"], p["ln_final.bias"]) @ p["text_projection"])
        return O.l2_normalize(torch.cat(feats))

    def run(fi, ft):
        p = {k: x.clone().requires_grad_(True) for k, x in state.items()}
        i, tt = fi(p), ft(p)
        loss = O.clip_loss(i, tt, p["logit_scale"].exp())
        loss = loss if torch.is_tensor(loss) else loss[0]
        loss.backward()
        return i.detach(), tt.detach(), {k: x.grad for k, x in p.items() if x.grad is not None}

    i0, t0, g0 = run(lambda p: O.encode_image(image, p, cfg), lambda p: O.encode_text(text, p, cfg))
    i1, t1, g1 = run(pruned_image, pruned_text)
    assert float((i1 - i0).abs().max()) < 1e-12 and float((t1 - t0).abs().max()) < 1e-12
    assert set(g0) == set(g1)
    worst = max(float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-300)) for k in g0)
    assert worst < 1e-10, worst
