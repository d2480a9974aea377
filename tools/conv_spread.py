import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import tests.test_gpu_convergence as T
import rnb_neus_fork_amd as R
z = np.load(os.path.join(T.GOLDEN, "convergence_ref.npz"), allow_pickle=False)
ref, alt = z["losses"], z["losses_alt"]
dec_ref, dec_alt = ref.reshape(10, -1).mean(axis=1), alt.reshape(10, -1).mean(axis=1)
lo = np.minimum(dec_ref, dec_alt) * (1.0 - T.ENVELOPE)
hi = np.maximum(dec_ref, dec_alt) * (1.0 + T.ENVELOPE)
# round 4: the default (forward-type products, weight gradients and FB as three fp16 terms), the round-3 arithmetic (six bf16
# terms everywhere), bf16 sweeps, and the deterministic forms of the fp32 and bf16 paths (ordered reductions instead of
# atomics: whatever spread is left there is not the atomics')
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for variant in ({}, {"x2h": False}, {"deterministic": True}, {"bf16": True}, {"bf16": True, "deterministic": True}):
    ps, pos, ml = [], [], []
    for i in range(RUNS):
        losses, psnr, mask_l1 = T._train(R, z, variant)
        dec = losses.reshape(10, -1).mean(axis=1)
        ps.append(psnr); ml.append(mask_l1)
        pos.append(float(np.max(np.maximum(lo - dec, dec - hi) / dec_ref)))
    print(variant, "PSNR", np.round(ps, 2).tolist(), "min", round(min(ps), 2), "max", round(max(ps), 2))
    print("   worst decile position", np.round(pos, 3).tolist(), "max", round(max(pos), 3))
    print("   mask_l1", np.round(ml, 4).tolist())
