"""Config C4, first phase: QuartzNet15x5 fine-tuning with the encoder frozen (what the reference's
FinetuneEncoderDecoder callback does until `unfreeze_encoder_at_epoch`), global batch 256 x 10 s split over the ranks.
Step = front end + frozen encoder forward + trainable decoder forward + CTC loss/gradient + decoder backward + the
gradient exchange (parallel.GradientSync: flat buffer, bf16 buckets launched from autograd hooks during backward, RCCL
reduce-scatter + all-gather) + fused AdamW.  A measurement tool, not the bench line:
    python tools/bench_finetune.py                      # 1 GPU, local batch 32
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_finetune.py --global-batch 256
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--global-batch", type=int, default=0, help="default: 32 per rank")
    ap.add_argument("--seconds", type=int, default=10)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--unfreeze", action="store_true", help="phase 2: encoder trainable (training-mode kernels, full backward)")
    ap.add_argument("--graph-encoder", action="store_true", help="phase 1: replay front end + frozen encoder from a hipGraph")
    ap.add_argument("--graph", action="store_true", help="phase 2: replay encoder + decoder + loss + backward from one hipGraph (train_graph.GraphedTrainStep)")
    ap.add_argument("--gemm-bf16", action="store_true", help="phase 2: bf16 operands for the pointwise-conv GEMMs (opt-in mixed precision)")
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from thunder_speech_amd.optim import FusedAdamW
    from thunder_speech_amd.parallel import GradientSync, max_over_ranks
    from thunder_speech_amd.quartznet.compatibility import build_synthetic_quartznet
    from thunder_speech_amd.utils import variance_preserving_init_
    torch.manual_seed(0)
    m = build_synthetic_quartznet(repeat_blocks=3)
    variance_preserving_init_(m.encoder, m.decoder, seed=0)
    m = m.to(dev)
    m.train()
    if args.gemm_bf16:
        from thunder_speech_amd import train_ops
        train_ops.set_gemm_precision("bf16")
    if not args.unfreeze:
        m.encoder.eval()
        for p in m.encoder.parameters():
            p.requires_grad_(False)
    if args.graph_encoder and not args.unfreeze:
        m.graph_frozen_encoder()
    trainable = [p for p in m.parameters() if p.requires_grad]
    opt = FusedAdamW(trainable, lr=1e-3)
    sync = GradientSync(trainable)
    B = (args.global_batch // world) if args.global_batch else 32
    g = torch.Generator().manual_seed(1234 + rank)
    wav = (0.1 * torch.randn(B, 16000 * args.seconds, generator=g)).to(dev)
    lengths = torch.full((B,), 16000.0 * args.seconds, device=dev)
    texts = ["".join(chr(97 + int(c)) for c in torch.randint(0, 26, (int(n),), generator=g)) for n in torch.randint(60, 140, (B,), generator=g)]

    graphed = None
    if args.graph and args.unfreeze:
        from thunder_speech_amd.train_graph import GraphedTrainStep
        graphed = GraphedTrainStep(m, opt, sync, max_target_len=160)

    def step():
        if graphed is not None:
            return graphed((wav, lengths, texts))
        sync.zero_grad()
        loss = m.training_step((wav, lengths, texts), 0)
        loss.backward()
        sync.finish()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = max_over_ranks((time.perf_counter() - t0) / args.steps, dev)
    if rank == 0:
        print(f"C4 phase {2 if args.unfreeze else 1} ({'encoder unfrozen' if args.unfreeze else 'frozen encoder'}{', bf16 activations' if args.gemm_bf16 else ''}{', hipGraph' if graphed is not None else ''}), {world} GPU(s), local batch {B} x {args.seconds} s: {dt * 1e3:.2f} ms/step, "
              f"{1 / dt:.1f} step/s, {world * B * args.seconds / dt:,.0f} audio-s/s, loss {float(loss.detach()):.3f}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
