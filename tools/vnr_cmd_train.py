#!/usr/bin/env python3
"""vnr_cmd_train with the reference's command line (apps/batch_trainer.cpp:30-141) on top of libvnr_amd:

  --volume <scene.json>      the ground truth volume (a VIDI3D / DIVA scene document)          [network.json]
  --network <model.json>     the neural network model configuration (comments allowed)          [network.json]
  --resume <params.json>     the pre-trained neural network (BSON)
  --report <file>            creating a training log file (csv: step,loss)                      [none]
  --max-num-steps <int>      maximum number of training steps                                   [1000]
  --training-mode / --mode   the data sampling mode: GPU | OUT_OF_CORE | NOTHING                [GPU]
  --quiet                    quiet mode
  --train-macrocell          train the macrocell grid at the same time

Like the reference it trains in bursts of 10 steps in fast mode, restarts when the loss is still > 0.9 after 5000 steps, prints
the Summary block (STEP / LOSS / TIME / PSNR / SSIM) and writes ./params.json (BSON).

More than one GPU (new: the reference is single-GPU): start one process per GPU with the torchrun environment (RANK, LOCAL_RANK,
WORLD_SIZE, MASTER_ADDR, MASTER_PORT; `python -m torch.distributed.run --nproc-per-node N tools/vnr_cmd_train.py ...` or any
launcher that sets them) and the steps become data-parallel steps on N x 65 536 samples (vnrAmdNeuralVolumeTrainDataParallel);
rank 0 reports and writes the files."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, dist  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser(description="Commandline Trainer")
    p.add_argument("--volume", default="network.json", metavar="filename", help="the ground truth volume")
    p.add_argument("--network", default="network.json", metavar="filename", help="the neural network model configuration")
    p.add_argument("--resume", default="", metavar="filename", help="the pre-trained neural network")
    p.add_argument("--report", default="none", metavar="filename", help="creating a trainning log file")
    p.add_argument("--max-num-steps", type=int, default=1000, metavar="int", help="maximum number of training steps")
    p.add_argument("--training-mode", "--mode", dest="training_mode", default="GPU", metavar="string", help="the data sampling mode")
    p.add_argument("--quiet", action="store_true", help="quiet mode")
    p.add_argument("--train-macrocell", action="store_true", help="train the macrocell grid at the same time")
    a = p.parse_args(argv)

    ctx = dist.init_from_env()        # one rank: binds the GPU; more: meets the other ranks (RCCL)
    root = ctx.rank == 0
    simple_volume = api.vnrCreateSimpleVolume(a.volume, a.training_mode)
    while True:
        # vnrCreateNeuralVolume(model, simple_volume, online_macrocell_construction = args.train_macrocell); a path is a model file
        neural_volume = api.vnrCreateNeuralVolume(a.network, simple_volume, bool(a.train_macrocell))
        if a.resume:
            api.vnrNeuralVolumeSetParams(neural_volume, a.resume)
        report = None
        if root and a.report not in ("", "none"):
            report = open(a.report if a.report.endswith(".csv") else a.report + ".csv", "w")
            report.write("step,loss\n")
        train_s, restart = 0.0, False
        for i in range(0, a.max_num_steps, 10):
            t0 = time.perf_counter()
            dist.train_data_parallel(ctx, neural_volume, 10, True)
            api.check(api.lib().vnrAmdSynchronize())
            train_s += time.perf_counter() - t0
            loss = api.vnrNeuralVolumeGetTrainingLoss(neural_volume)
            if report:
                report.write(f"{api.vnrNeuralVolumeGetTrainingStep(neural_volume)},{loss}\n")
            if root and not a.quiet:
                print(f"\r[train] {100.0 * i / max(a.max_num_steps, 1):5.1f} %  LOSS {loss:f}", end="", flush=True)
            if ctx.distributed:   # every rank takes the same decision: the largest loss any of them sees
                loss = dist.all_reduce_host([loss], dist.MAX)[0]
            if i >= 5000 and loss > 0.9:   # bad loss (batch_trainer.cpp:108-112)
                if root:
                    print("\nbad setup, ... restart")
                restart = True
                break
        if report:
            report.close()
        if not restart:
            break
    if root and not a.quiet:
        print()
    if not root:   # the replicas are identical: rank 0 evaluates and writes
        dist.barrier(ctx)
        dist.finalize()
        return 0
    psnr = api.vnrNeuralVolumeGetPSNR(neural_volume, a.report == "")
    ssim = api.vnrNeuralVolumeGetSSIM(neural_volume, a.report == "")
    print("Summary")
    print(f"  STEP={api.vnrNeuralVolumeGetTrainingStep(neural_volume)}")
    print(f"  LOSS={api.vnrNeuralVolumeGetTrainingLoss(neural_volume)}")
    print(f"  TIME={train_s}s")
    print(f"  PSNR={psnr}")
    print(f"  SSIM={ssim}")
    api.vnrNeuralVolumeSerializeParams(neural_volume, "params.json")
    if ctx.distributed:
        dist.barrier(ctx)
        dist.finalize()
    return 0


if __name__ == "__main__":
    sys.exit(main())
