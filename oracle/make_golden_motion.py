"""Generate tests/golden/recover_B3_T196.npz by running the UPSTREAM REFERENCE's post-sampling transform (build container
only; /root/reference imported read-only):

    python oracle/make_golden_motion.py

    sample/generate.py:163-166   inv_transform -> recover_from_ric(sample, 22) -> view/permute to [B, 22, 3, T]

The fixture stores the seeds and the reference's output; inputs are rebuilt from `motion_inputs` below.  The max-abs
difference between the reference and the restatement oracle/motion_oracle.py is merged into PIN_REPORT.json."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("MDM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def motion_inputs(B, T, seed, JF=263):
    """Seeded stand-ins for a sampler output and the dataset's Mean.npy / Std.npy (not shipped with the reference)."""
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(B, JF, 1, T, generator=g)
    mean = 0.3 * torch.randn(JF, generator=g)
    std = 0.05 + torch.rand(JF, generator=g)
    std[0] = 0.02 + 0.02 * std[0]           # rotation velocity: keep the heading angle within about a turn over 196
    mean[0] = 0.01 * mean[0]                # frames (as in real motions), so that cos/sin are not evaluated at ~100 rad
    return sample, mean, std


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from data_loaders.humanml.scripts.motion_process import recover_from_ric   # motion_process.py:437
    from oracle import motion_oracle as mo

    B, T, seed = 3, 196, 4242
    sample, mean, std = motion_inputs(B, T, seed)
    with torch.no_grad():
        x = (sample.permute(0, 2, 3, 1) * std + mean).float()                  # dataset.py:132-133, generate.py:163
        ref = recover_from_ric(x, 22)                                          # generate.py:164
        ref = ref.view(-1, *ref.shape[2:]).permute(0, 2, 3, 1).contiguous()    # generate.py:165
    got = mo.recover_from_ric(sample.numpy(), mean.numpy(), std.numpy(), 22)
    err = float(np.abs(got.astype(np.float64) - ref.numpy().astype(np.float64)).max())
    np.savez_compressed(os.path.join(OUT, "recover_B3_T196.npz"), B=B, T=T, seed=seed, joints=22, out=ref.numpy())
    rp = os.path.join(OUT, "PIN_REPORT.json")
    report = json.load(open(rp)) if os.path.isfile(rp) else {}
    # kept out of "cases" (those are absolute max-abs on O(1) samples): positions integrate to ~1e2, the pin is relative
    report["recover_B3_T196"] = {"oracle_vs_reference_maxabs": err, "out_absmax": float(np.abs(ref.numpy()).max()),
                                 "relative": err / float(np.abs(ref.numpy()).max())}
    report.get("cases", {}).pop("recover_B3_T196", None)
    json.dump(report, open(rp, "w"), indent=1, sort_keys=True)
    print("recover_B3_T196: oracle vs reference max-abs", err, "| |out|max", float(np.abs(ref.numpy()).max()))


if __name__ == "__main__":
    main()
