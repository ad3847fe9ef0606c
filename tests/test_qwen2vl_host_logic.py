"""CPU: the PRODUCT's Qwen2-VL host logic (mantis_amd/modeling_qwen2_vl.py: 3-D rope index, 2-D vision rotary ids, per-image attention
grouping, tower + merger sequencing, image-token merge, decoder with q/k/v bias and multimodal RoPE, gradient plumbing) with the
oracle's operator restatement monkeypatched in place of the HIP backend, against the Qwen2-VL oracle (pinned to the HF class the
reference resolves to) on the golden inputs."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh

CASES = ["qwen2vl_b1_img2", "qwen2vl_b1_img1_tall", "qwen2vl_b2_rightpad", "qwen2vl_b1_text_only"]


@pytest.fixture()
def cpu_backend(monkeypatch):
    import mantis_amd.modeling_qwen2_vl as mod
    from oracle import ops_ref
    monkeypatch.setattr(mod, "K", ops_ref)
    return mod


@pytest.mark.parametrize("case", CASES)
def test_step_matches_oracle(cpu_backend, case):
    z = Hh.load_case(case)
    model = Hh.build_qwen2vl_product("cpu")
    oracle = Hh.build_qwen2vl_oracle_bf16()
    assert model._ensure_grad_arena()
    rec = {}
    out = model.engine.step_from_batch(Hh.qwen2vl_batch(z), compute_grads=True, overwrite_grads=True, need_logits=True, record=rec)
    if "position_ids" in z.files:
        assert np.array_equal(rec["position_ids"].numpy(), z["position_ids"])        # vs the reference's own get_rope_index
    Hh.check_qwen2vl_step_against_oracle(model, oracle, z, out, rec)
    assert abs(float(out["loss"]) - float(z["loss"])) < 0.03 * float(z["loss"])      # vs the reference's own fp32 loss
    for n, p in model.named_parameters():
        if n.startswith("model.visual."):
            assert p.grad is None


def test_trainer_drives_the_qwen2vl_engine(cpu_backend):
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("qwen2vl_b2_rightpad")
    model = Hh.build_qwen2vl_product("cpu")
    tr = MantisHipTrainer(model, gradient_accumulation_steps=2)
    l1 = tr.training_step(model, Hh.qwen2vl_batch(z))
    g1 = model.grad_arena.float().clone()
    l2 = tr.training_step(model, Hh.qwen2vl_batch(z))
    assert l1.dim() == 0 and abs(float(l1) - float(z["loss"]) / 2) < 0.03 * float(z["loss"]) and torch.equal(l1, l2)
    assert Hh.rel_l2(model.grad_arena.float().numpy(), 2 * g1.numpy()) < 2e-2          # accumulation over the GA window


def test_image_token_count_mismatch_raises(cpu_backend):
    z = Hh.load_case("qwen2vl_b1_img2")
    model = Hh.build_qwen2vl_product("cpu")
    b = Hh.qwen2vl_batch(z)
    b["input_ids"] = b["input_ids"].clone()
    b["input_ids"][0, 4] = 5                      # one <|image_pad|> token fewer than merged patches
    with pytest.raises(ValueError):
        model.engine.step_from_batch(b, compute_grads=False)


def test_labelled_padding_is_rejected(cpu_backend):
    z = Hh.load_case("qwen2vl_b2_rightpad")
    model = Hh.build_qwen2vl_product("cpu")
    b = Hh.qwen2vl_batch(z)
    b["labels"] = b["labels"].clone()
    b["labels"][1, -1] = 7                        # a label on a padded position: HF would count it, this path refuses
    with pytest.raises(NotImplementedError):
        model.engine.step_from_batch(b, compute_grads=False)


def test_state_dict_names_match_reference_and_hf4_names_load():
    model = Hh.build_qwen2vl_product("cpu")
    _, sd = Hh.golden_cfg_and_weights("qwen2vl")
    assert set(dict(model.named_parameters())) == set(sd)
    # a transformers-4.x checkpoint (visual.*, model.layers.*) lands on the same parameters
    old = {}
    for k, v in sd.items():
        if k.startswith("model.visual."):
            old[k[len("model."):]] = v
        elif k.startswith("model.language_model."):
            old["model." + k[len("model.language_model."):]] = v
        else:
            old[k] = v
    m2 = Hh.build_qwen2vl_product("cpu")
    m2.arena.zero_()
    assert m2.load_reference_state_dict(old) == []
    assert torch.equal(m2.arena, model.arena)


def test_forward_contract(cpu_backend):
    z = Hh.load_case("qwen2vl_b1_img2")
    model = Hh.build_qwen2vl_product("cpu")
    model.eval()
    b = Hh.qwen2vl_batch(z)
    with torch.no_grad():
        out = model(**b)
    assert out.logits.dtype == torch.float32 and tuple(out.logits.shape) == tuple(z["logits"].shape)
    assert abs(float(out.loss) - float(z["loss"])) < 0.03 * float(z["loss"])
    assert out["loss"] is out.loss and out[0] is out.loss
    with pytest.raises(NotImplementedError):
        model(**b, pixel_values_videos=torch.zeros(1, 1))


# ----------------------------------------------------------------------------------------------------------- data parallel (gloo, 2 ranks)
def test_grad_buckets_tile_the_arena_and_every_bucket_is_signalled(cpu_backend):
    z = Hh.load_case("qwen2vl_b1_img2")
    for precision in ("bf16", "fp8"):
        model = Hh.build_qwen2vl_product("cpu").set_precision(precision)
        b = model.grad_buckets()
        spans = sorted((v.data_ptr(), v.numel()) for v in b.values())
        assert spans[0][0] == model.grad_arena.data_ptr() and sum(n for _, n in spans) == model.grad_arena.numel()
        for (p0, n0), (p1, _) in zip(spans, spans[1:]):
            assert p0 + 2 * n0 == p1, "buckets overlap or leave a gap"
        seen = []
        model.engine.step_from_batch(Hh.qwen2vl_batch(z), compute_grads=True, overwrite_grads=True, on_bucket_ready=seen.append)
        assert seen[0] == "head" and seen[-1] == "front" and set(seen) == set(b) and len(seen) == len(b), precision
        nl = model.config.text_config.num_hidden_layers
        assert seen[1:4] == [("layer", nl - 1, "down"), ("layer", nl - 1, "gu"), ("layer", nl - 1, "attn")]


def _dp_worker(rank, world, port, q, precision):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mantis_amd.modeling_qwen2_vl as mod
        from oracle import ops_ref
        mod.K = ops_ref
        from mantis_amd.trainer import MantisHipTrainer
        from mantis_amd.dp import GradReducer
        model = Hh.build_qwen2vl_product("cpu").set_precision(precision)
        z = Hh.load_case(["qwen2vl_b1_img2", "qwen2vl_b1_img1_tall"][rank])          # different samples (and lengths) per rank
        tr = MantisHipTrainer(model, gradient_accumulation_steps=1, reducer=GradReducer(model))
        loss = tr.training_step(model, Hh.qwen2vl_batch(z))
        q.put((rank, float(loss), model.grad_arena.float().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_dp2_gradients_are_the_mean_of_the_per_rank_gradients(cpu_backend, precision):
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, precision)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][2], res[1][2]), "ranks disagree after the bucketed all-reduce"
    acc = None
    for case in ("qwen2vl_b1_img2", "qwen2vl_b1_img1_tall"):
        model = Hh.build_qwen2vl_product("cpu").set_precision(precision)
        model._ensure_grad_arena()
        model.engine.step_from_batch(Hh.qwen2vl_batch(Hh.load_case(case)), compute_grads=True, overwrite_grads=True)
        g = model.grad_arena.float().numpy().copy()
        acc = g if acc is None else acc + g
    assert Hh.rel_l2(res[0][2], acc / 2) < 1e-2                                        # bf16 rounding of the averaged buckets


# ----------------------------------------------------------------------------------------------------------- sample packing
def _packed_from_b2(z):
    """The two samples of the B=2 golden batch (right padding dropped) as ONE packed row + segment ids; images in order of appearance."""
    ids, am, lab = z["input_ids"], z["attention_mask"], z["labels"]
    keep = [am[b].astype(bool) for b in range(2)]
    pid = torch.from_numpy(np.concatenate([ids[b][keep[b]] for b in range(2)]))[None]
    plab = torch.from_numpy(np.concatenate([lab[b][keep[b]] for b in range(2)]))[None]
    seg = torch.from_numpy(np.concatenate([np.full(int(keep[b].sum()), b, np.int32) for b in range(2)]))[None]
    return pid, plab, seg, torch.from_numpy(z["pixel_values"]), torch.from_numpy(z["image_grid_thw"])


@pytest.mark.parametrize("precision", ["bf16", "fp8"])
def test_packed_row_equals_the_separate_samples(cpu_backend, precision):
    from oracle.qwen2vl_ref import rope_index
    z = Hh.load_case("qwen2vl_b2_rightpad")
    pid, plab, seg, pv, grid = _packed_from_b2(z)
    model = Hh.build_qwen2vl_product("cpu").set_precision(precision)
    oracle = Hh.build_qwen2vl_oracle_bf16()
    assert model._ensure_grad_arena()
    rec = {}
    out = model.engine.step(pid, torch.ones_like(pid), plab, pv, grid, compute_grads=True, overwrite_grads=True, segment_ids=seg, record=rec)
    # the 3-D rope index of the packed row = each sample's own index (the reference's get_rope_index on the sample alone), side by side
    n0 = int((seg[0] == 0).sum())
    p0 = rope_index(pid[:, :n0], None, grid[:2], model.config.image_token_id, 2)
    p1 = rope_index(pid[:, n0:], None, grid[2:], model.config.image_token_id, 2)
    assert torch.equal(rec["position_ids"], torch.cat([p0, p1], dim=2))
    oracle.zero_grad()
    oloss = oracle.forward_packed(pid, pv, grid, seg, plab)
    oloss.backward()
    fp8 = precision == "fp8"
    assert abs(float(out["loss"]) - float(oloss)) <= (1e-2 if fp8 else 5e-3) * float(oloss)
    for name, p in model.named_parameters():
        if p.requires_grad:
            g, og = p.grad.float().numpy(), oracle.w[name].grad.numpy()
            c = Hh.cosine(g, og)
            assert c > ((0.85 if p.dim() == 1 else 0.95) if fp8 else 0.995), (name, c)


def test_loss_backward_through_the_autograd_bridge(cpu_backend):
    """Stock `model(**batch).loss.backward()` callers (HF Trainer.training_step without the subclass): same loss and bit-identical
    gradients as MantisHipTrainer.training_step; a second backward after zero_grad overwrites, without one accumulates."""
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("qwen2vl_b2_rightpad")
    ref = Hh.build_qwen2vl_product("cpu")
    l_ref = MantisHipTrainer(ref, gradient_accumulation_steps=1).training_step(ref, Hh.qwen2vl_batch(z))
    model = Hh.build_qwen2vl_product("cpu")
    model.train()
    out = model(**Hh.qwen2vl_batch(z))
    assert out.logits is None and out.loss.requires_grad
    out.loss.backward()
    assert torch.equal(out.loss.detach(), l_ref) and torch.equal(model.grad_arena, ref.grad_arena)
    out2 = model(**Hh.qwen2vl_batch(z))
    (out2.loss / 2).backward()                                      # no zero_grad in between: accumulates, with the caller's scale
    assert Hh.rel_l2(model.grad_arena.float().numpy(), 1.5 * ref.grad_arena.float().numpy()) < 1e-2
    for p in model.parameters():
        p.grad = None
    model(**Hh.qwen2vl_batch(z)).loss.backward()
    assert torch.equal(model.grad_arena, ref.grad_arena)


def test_pack_samples_feeds_the_packed_qwen2vl_step(cpu_backend):
    """data.pack_samples on two Qwen2-VL samples (patches and image_grid_thw concatenated in order of appearance) + the trainer's
    packed-batch route (4-D block-diagonal mask or segment ids) = the packed step of test_packed_row_equals_the_separate_samples."""
    from mantis_amd.data import pack_samples
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("qwen2vl_b2_rightpad")
    ids, am, lab = z["input_ids"], z["attention_mask"], z["labels"]
    grids, pv = z["image_grid_thw"], z["pixel_values"]
    n0 = int((grids[:2, 0] * grids[:2, 1] * grids[:2, 2]).sum())
    keep = [am[b].astype(bool) for b in range(2)]
    samples = [dict(input_ids=ids[0][keep[0]], labels=lab[0][keep[0]], pixel_values=pv[:n0], image_grid_thw=grids[:2]),
               dict(input_ids=ids[1][keep[1]], labels=lab[1][keep[1]], pixel_values=pv[n0:], image_grid_thw=grids[2:])]
    packed = pack_samples(samples)
    assert torch.equal(packed["image_grid_thw"], torch.from_numpy(grids)) and torch.equal(packed["pixel_values"], torch.from_numpy(pv))
    pid, plab, seg, pv_t, grid_t = _packed_from_b2(z)
    assert torch.equal(packed["input_ids"], pid) and torch.equal(packed["segment_ids"], seg)
    ref = Hh.build_qwen2vl_product("cpu")
    ref._ensure_grad_arena()
    out = ref.engine.step(pid, torch.ones_like(pid), plab, pv_t, grid_t, compute_grads=True, overwrite_grads=True, segment_ids=seg)
    model = Hh.build_qwen2vl_product("cpu")
    loss = MantisHipTrainer(model, gradient_accumulation_steps=1).training_step(model, packed)
    assert torch.equal(loss, out["loss"].reshape(())) and torch.equal(model.grad_arena, ref.grad_arena)


def test_rope_index_random_layouts_match_the_oracle():
    """The product's host-side get_rope_index (modeling_qwen2_vl.mrope_position_ids) against the oracle's restatement (pinned to the HF
    output on the goldens) on random batches: several images per row, left / right padding, text-only rows; and the tower's 2-D ids."""
    from mantis_amd.modeling_qwen2_vl import mrope_position_ids, vision_hw_ids
    from oracle.qwen2vl_ref import rope_index, vision_position_ids
    rng = np.random.default_rng(5)
    IMG = 7
    for trial in range(60):
        B = int(rng.integers(1, 4))
        rows, grids = [], []
        for b in range(B):
            toks = []
            for _ in range(int(rng.integers(0, 4))):
                toks += rng.integers(10, 50, size=int(rng.integers(1, 6))).tolist()      # at least one text token between two images
                h, w = 2 * int(rng.integers(1, 4)), 2 * int(rng.integers(1, 4))
                grids.append((1, h, w))
                toks += [IMG] * (h * w // 4)
            toks += rng.integers(10, 50, size=int(rng.integers(1, 6))).tolist()
            rows.append(toks)
        T = max(len(r) for r in rows)
        ids = np.zeros((B, T), np.int64)
        am = np.zeros((B, T), np.int64)
        for b, r in enumerate(rows):
            off = (T - len(r)) if (trial % 2) else 0            # left padding on odd trials, right padding on even ones
            ids[b, off: off + len(r)] = r
            am[b, off: off + len(r)] = 1
        want = rope_index(ids, am, np.array(grids, np.int64).reshape(-1, 3), IMG, 2)
        got = mrope_position_ids(torch.from_numpy(ids), torch.from_numpy(am), grids, IMG, 2)
        assert torch.equal(got, want), trial
    g = [(1, 4, 6), (2, 2, 4)]
    assert torch.equal(vision_hw_ids(g, 2).t(), vision_position_ids(np.array(g), 2))


@pytest.mark.parametrize("precision", ["bf16", "fp8", "fp8_rowwise"])
def test_activation_checkpointing_is_bit_identical(cpu_backend, precision):
    """gradient_checkpointing_enable() on the Qwen2-VL path (multimodal RoPE tables handed to the shared decoder loop; with fp8 linears the
    layer's quantisations run again too -- amax is a maximum, so the scales come out the same): same loss, same gradients."""
    z = Hh.load_case(CASES[0])
    res = []
    for on in (False, True):
        model = Hh.build_qwen2vl_product("cpu")
        if precision != "bf16":
            model.set_precision(precision)
        if on:
            model.gradient_checkpointing_enable()
        model._ensure_grad_arena()
        out = model.engine.step_from_batch(Hh.qwen2vl_batch(z), compute_grads=True, overwrite_grads=True)
        res.append((float(out["loss"]), model.grad_arena.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]) and float(res[0][1].float().abs().sum()) > 0
