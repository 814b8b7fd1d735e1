"""Device-side post-sampling transform: what sample/generate.py:160-166 does on the CPU after a
D2H copy -- `inv_transform` (data_loaders/humanml/data/dataset.py:132-133) followed by `recover_from_ric`
(data_loaders/humanml/scripts/motion_process.py:437-452) and the permute to [B, joints, 3, T] -- as one HIP kernel
on the sampler's output tensor (csrc/motion_recover.h behind mdm_recover_from_ric).  No CPU fallback."""
import torch

from . import _native as nat


def recover_from_ric(sample, mean, std, joints_num=None, _native_lib=None):
    """sample: float32 [B, njoints_feat, 1, T] (the output of p_sample_loop, normalised features);
    mean/std: float32 [njoints_feat] (the dataset's Mean.npy / Std.npy);  returns float32 [B, joints_num, 3, T],
    i.e. generate.py's `recover_from_ric(inv_transform(sample.permute(0, 2, 3, 1)), n_joints)` reshaped as at :166."""
    lib = _native_lib if _native_lib is not None else nat.load_native()
    emulation = not lib.path.endswith(nat.LIB_NAME)
    if sample.dim() != 4 or sample.shape[2] != 1:
        raise ValueError(f"sample must be [B, njoints_feat, 1, T], got {tuple(sample.shape)}")
    if not emulation and not sample.is_cuda:
        raise nat.MdmError("the MI355X HIP path needs tensors on a cuda (ROCm) device; got " + str(sample.device))
    B, JF, _, T = sample.shape
    if joints_num is None:
        joints_num = 22 if JF == 263 else 21            # generate.py:162
    x = sample.contiguous().float()
    mean = torch.as_tensor(mean, dtype=torch.float32, device=x.device).contiguous()
    std = torch.as_tensor(std, dtype=torch.float32, device=x.device).contiguous()
    if mean.numel() != JF or std.numel() != JF:
        raise ValueError("mean/std must have njoints_feat elements")
    out = torch.empty(B, joints_num, 3, T, dtype=torch.float32, device=x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else None
    lib.check(lib.mdm_recover_from_ric(x.data_ptr(), mean.data_ptr(), std.data_ptr(), out.data_ptr(), B, T, JF,
                                       joints_num, stream), "mdm_recover_from_ric")
    return out
