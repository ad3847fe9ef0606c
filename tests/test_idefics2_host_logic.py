"""CPU: the PRODUCT's Idefics2 host logic (mantis_amd/modeling_idefics2.py: image preparation, vision tower sequencing, connector
forward / backward, merger, decoder, gradient plumbing) with the oracle's operator restatement monkeypatched in place of the HIP
backend, against the Idefics2 oracle (pinned to the reference fork) on the golden inputs."""
import numpy as np
import pytest
import torch

from tests import helpers as Hh

CASES = ["idefics2_b1_img2", "idefics2_b1_navit", "idefics2_b2_padimg_rightpad", "idefics2_b1_text_only"]


@pytest.fixture()
def cpu_backend(monkeypatch):
    import mantis_amd.modeling_idefics2 as mod
    from oracle import ops_ref
    monkeypatch.setattr(mod, "K", ops_ref)
    return mod


@pytest.mark.parametrize("case", CASES)
def test_step_matches_oracle(cpu_backend, case):
    z = Hh.load_case(case)
    model = Hh.build_idefics2_product("cpu")
    oracle = Hh.build_idefics2_oracle_bf16()
    assert model._ensure_grad_arena()
    rec = {}
    out = model.engine.step_from_batch(Hh.idefics2_batch(z), compute_grads=True, overwrite_grads=True, need_logits=True, record=rec)
    Hh.check_idefics2_step_against_oracle(model, oracle, z, out, rec)
    assert abs(float(out["loss"]) - float(z["loss"])) < 0.03 * float(z["loss"])          # vs the reference's own fp32 loss
    for n, p in model.named_parameters():
        if n.startswith("model.vision_model."):
            assert p.grad is None


def test_trainer_drives_the_idefics2_engine(cpu_backend):
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("idefics2_b2_padimg_rightpad")
    model = Hh.build_idefics2_product("cpu")
    tr = MantisHipTrainer(model, gradient_accumulation_steps=2)
    l1 = tr.training_step(model, Hh.idefics2_batch(z))
    g1 = model.grad_arena.float().clone()
    l2 = tr.training_step(model, Hh.idefics2_batch(z))
    assert l1.dim() == 0 and abs(float(l1) - float(z["loss"]) / 2) < 0.03 * float(z["loss"]) and torch.equal(l1, l2)
    assert Hh.rel_l2(model.grad_arena.float().numpy(), 2 * g1.numpy()) < 2e-2          # accumulation over the GA window


def test_image_token_count_mismatch_raises(cpu_backend):
    z = Hh.load_case("idefics2_b1_img2")
    model = Hh.build_idefics2_product("cpu")
    b = Hh.idefics2_batch(z)
    b["input_ids"] = b["input_ids"].clone()
    b["input_ids"][0, 3] = 5                      # one <image> token fewer than image hidden states
    with pytest.raises(ValueError):
        model.engine.step_from_batch(b, compute_grads=False)


def test_state_dict_names_match_reference():
    model = Hh.build_idefics2_product("cpu")
    _, sd = Hh.golden_cfg_and_weights("idefics2")
    assert {n for n, _ in model.named_parameters()} == set(sd)
    assert sum(b.numel() for b in model.grad_buckets().values()) == model.grad_arena.numel()


def test_packed_row_equals_the_separate_samples(cpu_backend):
    """Long-sequence packing on the Idefics2 path (BASELINE configs[3]): the two samples of the B=2 golden batch (unpadded parts)
    packed into one row -- loss and gradients equal the oracle running them one by one."""
    z = Hh.load_case("idefics2_b2_padimg_rightpad")
    ids, am, lab = z["input_ids"], z["attention_mask"], z["labels"]
    keep = [am[b].astype(bool) for b in range(2)]
    pid = torch.from_numpy(np.concatenate([ids[b][keep[b]] for b in range(2)]))[None]
    plab = torch.from_numpy(np.concatenate([lab[b][keep[b]] for b in range(2)]))[None]
    seg = torch.from_numpy(np.concatenate([np.full(int(keep[b].sum()), b, np.int32) for b in range(2)]))[None]
    pv = torch.from_numpy(np.concatenate([z["pixel_values"][0], z["pixel_values"][1][:1]], 0))[None]        # 3 real images in order
    pm = torch.from_numpy(np.concatenate([z["pixel_attention_mask"][0], z["pixel_attention_mask"][1][:1]], 0))[None]
    model = Hh.build_idefics2_product("cpu")
    oracle = Hh.build_idefics2_oracle_bf16()
    assert model._ensure_grad_arena()
    out = model.engine.step(pid, torch.ones_like(pid), plab, pv, pm, compute_grads=True, overwrite_grads=True, segment_ids=seg)
    oracle.zero_grad()
    oloss = oracle.forward_packed(pid, pv, pm, seg, torch.ones_like(pid), plab)
    oloss.backward()
    assert abs(float(out["loss"]) - float(oloss)) <= 5e-3 * float(oloss), (float(out["loss"]), float(oloss))
    for name, p in model.named_parameters():
        if p.requires_grad:
            g, og = p.grad.float().numpy(), oracle.w[name].grad.numpy()
            assert Hh.cosine(g, og) > 0.995 and Hh.rel_l2(g, og) < 6e-2, (name, Hh.cosine(g, og), Hh.rel_l2(g, og))


def test_loss_backward_through_the_autograd_bridge(cpu_backend):
    """Stock `model(**batch).loss.backward()` callers: same loss and bit-identical gradients as MantisHipTrainer.training_step."""
    import torch
    from mantis_amd.trainer import MantisHipTrainer
    z = Hh.load_case("idefics2_b2_padimg_rightpad")
    ref = Hh.build_idefics2_product("cpu")
    l_ref = MantisHipTrainer(ref, gradient_accumulation_steps=1).training_step(ref, Hh.idefics2_batch(z))
    model = Hh.build_idefics2_product("cpu")
    model.train()
    out = model(**Hh.idefics2_batch(z))
    assert out.logits is None and out.loss.requires_grad
    out.loss.backward()
    assert torch.equal(out.loss.detach(), l_ref) and torch.equal(model.grad_arena, ref.grad_arena)
    for p in model.parameters():
        p.grad = None
    model(**Hh.idefics2_batch(z)).loss.backward()
    assert torch.equal(model.grad_arena, ref.grad_arena)


# ----------------------------------------------------------------------------------------------------------- data parallel (gloo, 2 ranks)
def test_every_grad_bucket_is_signalled_once_in_backward_order(cpu_backend):
    z = Hh.load_case("idefics2_b1_img2")
    model = Hh.build_idefics2_product("cpu")
    b = model.grad_buckets()
    spans = sorted((v.data_ptr(), v.numel()) for v in b.values())
    assert spans[0][0] == model.grad_arena.data_ptr() and sum(n for _, n in spans) == model.grad_arena.numel()
    for (p0, n0), (p1, _) in zip(spans, spans[1:]):
        assert p0 + 2 * n0 == p1, "buckets overlap or leave a gap"
    seen = []
    model.engine.step_from_batch(Hh.idefics2_batch(z), compute_grads=True, overwrite_grads=True, on_bucket_ready=seen.append)
    assert set(seen) == set(b) and len(seen) == len(b)
    nl = model.config.text_config.num_hidden_layers
    layer_order = [s[1] for s in seen if isinstance(s, tuple) and s[0] == "layer" and s[2] == "down"]
    assert layer_order == list(range(nl - 1, -1, -1))                       # the decoder's buckets leave in backward order


def _dp_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mantis_amd.modeling_idefics2 as mod
        from oracle import ops_ref
        mod.K = ops_ref
        from mantis_amd.trainer import MantisHipTrainer
        from mantis_amd.dp import GradReducer
        model = Hh.build_idefics2_product("cpu")
        z = Hh.load_case(["idefics2_b1_img2", "idefics2_b1_navit"][rank])              # different samples (and image shapes) per rank
        tr = MantisHipTrainer(model, gradient_accumulation_steps=1, reducer=GradReducer(model))
        loss = tr.training_step(model, Hh.idefics2_batch(z))
        q.put((rank, float(loss), model.grad_arena.float().numpy()))
    finally:
        dist.destroy_process_group()


def test_dp2_gradients_are_the_mean_of_the_per_rank_gradients(cpu_backend):
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][2], res[1][2]), "ranks disagree after the bucketed all-reduce"
    acc = None
    for case in ("idefics2_b1_img2", "idefics2_b1_navit"):
        model = Hh.build_idefics2_product("cpu")
        model._ensure_grad_arena()
        model.engine.step_from_batch(Hh.idefics2_batch(Hh.load_case(case)), compute_grads=True, overwrite_grads=True)
        g = model.grad_arena.float().numpy().copy()
        acc = g if acc is None else acc + g
    assert Hh.rel_l2(res[0][2], acc / 2) < 1e-2                                        # bf16 rounding of the averaged buckets


def test_bucket_table_reproduces_the_reference_position_ids():
    """The device kernel (csrc/vit.hip navit_prepare_kernel) takes the NaViT position ids from a host-built table:
    pos[k-th attended patch] = tab[nh][k // nw] * side + tab[nw][k % nw].  The table + formula must equal the reference's per-image float
    arithmetic (oracle/idefics2_ref.bucketized_position_ids, modeling_idefics2.py:190-210) for every (nh, nw) up to the largest grid."""
    from oracle.idefics2_ref import bucketized_position_ids
    model = Hh.build_idefics2_product("cpu")
    vc = model.config.vision_config
    side = vc.image_size // vc.patch_size
    big = max(side, 9)
    tab = model.engine._bucket_table(big + 1, "cpu")
    for nh in range(1, big + 1):
        for nw in range(1, big + 1):
            pm = torch.zeros(1, big, big, dtype=torch.bool)
            pm[0, :nh, :nw] = True
            ref = bucketized_position_ids(pm, side)[0]
            k = torch.arange(nh * nw)
            got = torch.zeros(big * big, dtype=torch.int64)
            got[pm[0].reshape(-1)] = (tab[nh][k // nw].long() * side + tab[nw][k % nw].long())
            assert torch.equal(got, ref), (nh, nw)


def test_activation_checkpointing_is_bit_identical(cpu_backend):
    """gradient_checkpointing_enable() on the Idefics2 path (Mistral decoder through the shared decoder.py loop): same loss, same gradients."""
    z = Hh.load_case(CASES[0])
    res = []
    for on in (False, True):
        model = Hh.build_idefics2_product("cpu")
        if on:
            model.gradient_checkpointing_enable()
        model._ensure_grad_arena()
        out = model.engine.step_from_batch(Hh.idefics2_batch(z), compute_grads=True, overwrite_grads=True)
        res.append((float(out["loss"]), model.grad_arena.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
