"""Factory seam: the pieces of utils/model_util.py (:8-15, :18-21, :24-71, :75-116, :118-132) that
`sample/generate.py` goes through to obtain (model, diffusion), rebuilt on the MI355X classes.

Only what the sampling hot path needs is honoured: HumanML3D / KIT `hml_vec` shapes and the 25 x 6 rot6d
features of the action datasets, START_X + FIXED_SMALL/LARGE, no respacing.  Everything else raises in the class constructors.
"""
from types import SimpleNamespace

import torch

from . import gaussian_diffusion as gd
from .mdm import MDM
from .respace import SpacedDiffusion, space_timesteps

# utils/model_util.py:63-64 pins these; they are not argparse flags in the reference
FF_SIZE, NUM_HEADS, DROPOUT, ACTIVATION, CLIP_VERSION = 1024, 4, 0.1, "gelu", "ViT-B/32"
_POSE_DIMS = {"humanml": 263, "kit": 251}      # utils/model_util.py:41-49
# data_loaders/humanml_utils.py:30 + utils/model_util.py:45: the goal joints of a --multi_target_cond (target-conditioned DiP) checkpoint
HML_GOAL_JOINT_NAMES = ["pelvis", "left_foot", "right_foot", "left_wrist", "right_wrist", "head"]


def default_args(**over):
    """The argparse defaults that reach the hot path (utils/parser_util.py:74-130, :207) as a namespace."""
    a = dict(dataset="humanml", latent_dim=512, layers=8, arch="trans_enc", emb_trans_dec=False,
             cond_mask_prob=0.1, text_encoder_type="clip", pos_embed_max_len=5000, mask_frames=True,
             unconstrained=False, diffusion_steps=50, noise_schedule="cosine", sigma_small=True,
             lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0, pred_len=0, context_len=0, guidance_param=2.5)
    a.update(over)
    return SimpleNamespace(**a)


def get_cond_mode(args):
    """utils/parser_util.py:269-276."""
    if getattr(args, "unconstrained", False):
        return "no_cond"
    return "text" if args.dataset in ("kit", "humanml") else "action"


def get_model_args(args, data=None):
    """utils/model_util.py:24-71.  The hml_vec datasets (263 / 251 features) or, for the action datasets (humanact12 / uestc), the
    SMPL defaults of :33-37: 25 joints x 6 rot6d features, `num_actions` from the dataset object (:29-32; args.num_actions when no
    dataset object is at hand).  SMPL forward kinematics for rendering those (rot2xyz) stays outside the hot path."""
    g = vars(args).get
    if args.dataset in _POSE_DIMS:
        njoints, nfeats, data_rep, num_actions = _POSE_DIMS[args.dataset], 1, "hml_vec", 1
    else:
        njoints, nfeats, data_rep = 25, 6, "rot6d"
        num_actions = getattr(getattr(data, "dataset", None), "num_actions", g("num_actions", 1))
    return dict(modeltype="", njoints=njoints, nfeats=nfeats, num_actions=num_actions, translation=True,
                pose_rep="rot6d", glob=True, glob_rot=True, latent_dim=args.latent_dim, ff_size=FF_SIZE,
                num_layers=args.layers, num_heads=NUM_HEADS, dropout=DROPOUT, activation=ACTIVATION,
                data_rep=data_rep, cond_mode=get_cond_mode(args), cond_mask_prob=args.cond_mask_prob,
                action_emb="tensor", arch=args.arch, emb_trans_dec=g("emb_trans_dec", False),
                clip_version=CLIP_VERSION, dataset=args.dataset, text_encoder_type=g("text_encoder_type", "clip"),
                pos_embed_max_len=g("pos_embed_max_len", 5000), mask_frames=g("mask_frames", False),
                pred_len=g("pred_len", 0), context_len=g("context_len", 0), emb_policy=g("emb_policy", "add"),
                all_goal_joint_names=list(HML_GOAL_JOINT_NAMES) if args.dataset == "humanml" else [],
                multi_target_cond=g("multi_target_cond", False), multi_encoder_type=g("multi_encoder_type", "multi"),
                target_enc_layers=g("target_enc_layers", 1))


def create_gaussian_diffusion(args):
    """utils/model_util.py:75-116: x0-prediction, no learned sigma, no respacing."""
    steps = args.diffusion_steps
    var = gd.ModelVarType.FIXED_SMALL if args.sigma_small else gd.ModelVarType.FIXED_LARGE
    g = vars(args).get
    return SpacedDiffusion(use_timesteps=space_timesteps(steps, [steps]),
                           betas=gd.get_named_beta_schedule(args.noise_schedule, steps, 1.0),
                           model_mean_type=gd.ModelMeanType.START_X, model_var_type=var,
                           loss_type=gd.LossType.MSE, rescale_timesteps=False,
                           lambda_vel=g("lambda_vel", 0.0), lambda_rcxyz=g("lambda_rcxyz", 0.0),
                           lambda_fc=g("lambda_fc", 0.0), lambda_target_loc=g("lambda_target_loc", 0.0))


def create_model_and_diffusion(args, data=None, **model_over):
    return MDM(**{**get_model_args(args, data), **model_over}), create_gaussian_diffusion(args)


def load_model_wo_clip(model, state_dict):
    """utils/model_util.py:8-15: positional tables are recomputed, CLIP weights are not in checkpoints."""
    sd = dict(state_dict)
    sd.pop("sequence_pos_encoder.pe", None)
    sd.pop("embed_timestep.sequence_pos_encoder.pe", None)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert len(unexpected) == 0, unexpected
    assert all(k.startswith("clip_model.") or "sequence_pos_encoder" in k for k in missing), missing


def load_saved_model(model, model_path, use_avg=False):
    """utils/model_util.py:118-132."""
    ck = torch.load(model_path, map_location="cpu")
    if use_avg and "model_avg" in ck:
        ck = ck["model_avg"]
    elif "model" in ck:
        ck = ck["model"]
    load_model_wo_clip(model, ck)
    return model
