"""Worker of gpu_checks.check_dp_world2_on_gpu: one of TWO ranks that share the box's single MI355X.  RCCL refuses two ranks on one device,
so the transport is gloo (device tensors staged through the host by torch) -- everything else is the product's N > 1 path on the real
kernels: `GradReducer` signalled per bucket from inside the HIP backward, the mean landing in `finish()`, the gradient norm of the
REDUCED gradient (no fold into the dW epilogues under data parallelism), clip + FusedAdamW on every rank.
    python tests/dp_world2_gpu_worker.py <rank> <world> <port> <family> <golden case> <out.pt>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(family, dev):
    """(model, batch builder) of a model family on its golden weights: llava | idefics2 | qwen2vl | qwen2vl_fp8"""
    import torch
    from tests import helpers as Hh
    if family == "llava":
        model = Hh.build_product_model("siglip", dev)[0]
        return model, lambda z: dict(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                                     labels=torch.from_numpy(z["labels"]), pixel_values=Hh.pixels_list(z))
    if family == "idefics2":
        return Hh.build_idefics2_product(dev), Hh.idefics2_batch
    if family in ("qwen2vl", "qwen2vl_fp8"):
        model = Hh.build_qwen2vl_product(dev)
        return (model.set_precision("fp8") if family.endswith("fp8") else model), Hh.qwen2vl_batch
    raise ValueError(family)


def main():
    rank, world, port, family, case, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import mantis_amd  # noqa: F401  (GPU_MAX_HW_QUEUES before the first GPU call)
    import torch
    import torch.distributed as dist
    from tests import helpers as Hh
    from mantis_amd.dp import GradReducer
    from mantis_amd.optim import FusedAdamW
    from mantis_amd.trainer import MantisHipTrainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        model, make_batch = build(family, dev)
        red = GradReducer(model)
        assert red.active and red.world == world
        tr = MantisHipTrainer(model, gradient_accumulation_steps=1, reducer=red)
        opt = FusedAdamW(model, lr=1e-3, weight_decay=0.0, max_grad_norm=1.0)
        loss = tr.training_step(model, make_batch(Hh.load_case(case)))
        torch.cuda.synchronize()
        grads = model.grad_arena.detach().float().cpu().clone()
        opt.step()
        torch.cuda.synchronize()
        torch.save(dict(rank=rank, loss=float(loss), grads=grads, params=model.arena.detach().float().cpu().clone(),
                        grad_norm=None if opt.last_grad_norm is None else float(opt.last_grad_norm),
                        buckets=red.stats["buckets"], bytes=red.stats["bytes"]), out)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
