"""`MDM` -- the reference's denoiser seam (model/mdm.py:11-293) on the MI355X HIP path.

Same constructor keywords, same `forward(x, timesteps, y)` signature and `y` contract, same state-dict
keys, so `utils/model_util.py:18-21 create_model_and_diffusion` and
`load_saved_model` work against it unchanged.  The arithmetic runs in csrc/libmdm_hip.so; the torch
sub-modules below exist only to hold parameters under the reference's names.

Scope: arch='trans_enc', text conditioning with a cached `y['text_embed']`
(or no conditioning), inference only; and (DiP) arch='trans_dec' with prefix completion
and a cached token-level text embedding (DistilBERT) or a single CLIP token as the decoder memory (model/mdm.py:261-262; pinned
against the reference by tests/golden/dip_clip_*.npz since round 4).  In the default f16x3 mode the decoder stack runs on fp16
hi/lo operand planes like the encoder (small-tile split-precision GEMMs with all three LayerNorms of a layer folded, split-precision
self-attention -- with or without a frame mask, DiP.md:181 trains with --mask_frames); `precision='f32'` is exact fp32 MFMA.
Everything else raises NotImplementedError loudly.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import _native as nat
from . import _engine
from ._engine import Engine


class PositionalEncoding(nn.Module):
    """model/mdm.py:296-313 -- the table is built on the host exactly as the reference builds it."""

    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0).transpose(0, 1))   # [max_len, 1, d]


class TimestepEmbedder(nn.Module):
    """model/mdm.py:316-330 (parameters only; evaluated as a precomputed table inside the library)."""

    def __init__(self, latent_dim, sequence_pos_encoder):
        super().__init__()
        self.latent_dim = latent_dim
        self.sequence_pos_encoder = sequence_pos_encoder
        self.time_embed = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.SiLU(), nn.Linear(latent_dim, latent_dim))


class InputProcess(nn.Module):
    """model/mdm.py:333-357 (parameters only)."""

    def __init__(self, data_rep, input_feats, latent_dim):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.poseEmbedding = nn.Linear(input_feats, latent_dim)


class OutputProcess(nn.Module):
    """model/mdm.py:360-386 (parameters only)."""

    def __init__(self, data_rep, input_feats, latent_dim, njoints, nfeats):
        super().__init__()
        self.data_rep, self.input_feats, self.latent_dim = data_rep, input_feats, latent_dim
        self.njoints, self.nfeats = njoints, nfeats
        self.poseFinal = nn.Linear(latent_dim, input_feats)


class EmbedAction(nn.Module):
    """model/mdm.py:389-397: one learned row per action class (`cond_mode='action'`, the humanact12 / uestc checkpoints)."""

    def __init__(self, num_actions, latent_dim):
        super().__init__()
        self.action_embedding = nn.Parameter(torch.randn(num_actions, latent_dim))

    def forward(self, input):
        return self.action_embedding[input[:, 0].to(torch.long)]


def _torch_linear(h, weight, bias, silu):
    h = torch.nn.functional.linear(h, weight, bias)
    return h * torch.sigmoid(h) if silu else h


class _TargetLocBase(nn.Module):
    """What the three target-location encoders of model/mdm.py:399-479 share: the joint list (`all_goal_joint_names` + 'traj' +
    'heading'), the per-sample choice of joints -- here one [B, n_ext] 0/1 matrix instead of the reference's loops -- and the
    Linear (SiLU Linear)* stacks.  `linear(h, weight, bias, silu)` is the dense layer the stacks run on: torch by default (a module
    called on its own, as in the reference); MDM.target_embedding passes the library's own fp32 GEMM (Engine.linear), whose k order
    does not depend on the row count, so that a sample's embedding is the same bits in whatever batch or shard it is evaluated."""

    def __init__(self, all_goal_joint_names, latent_dim):
        super().__init__()
        self.extended_goal_joint_names = list(all_goal_joint_names) + ['traj', 'heading']
        self.latent_dim = latent_dim

    def chosen(self, input, target_joint_names, target_heading):
        sel = torch.zeros(input.shape[0], len(self.extended_goal_joint_names))
        for b, names in enumerate(target_joint_names):
            for j in list(names) + (['heading'] if target_heading[b] else []):
                sel[b, self.extended_goal_joint_names.index(str(j))] = 1.0
        return sel.to(device=input.device, dtype=input.dtype)

    @staticmethod
    def _mlp(n_in, width, num_layers):
        layers = [nn.Linear(n_in, width)]
        for _ in range(num_layers):
            layers += [nn.SiLU(), nn.Linear(width, width)]
        return nn.Sequential(*layers)

    @staticmethod
    def _run(seq, h, linear):
        lins = [m for m in seq if isinstance(m, nn.Linear)]
        for i, lin in enumerate(lins):
            h = linear(h.contiguous(), lin.weight, lin.bias, i + 1 < len(lins))
        return h


class EmbedTargetLocSingle(_TargetLocBase):
    """model/mdm.py:399-419 (`--multi_encoder_type single`): ONE MLP over every joint's (x, y, z, chosen) numbers."""

    def __init__(self, all_goal_joint_names, latent_dim, num_layers=1):
        super().__init__(all_goal_joint_names, latent_dim)
        self.target_cond_dim = 4 * len(self.extended_goal_joint_names)
        self.mlp = self._mlp(self.target_cond_dim, latent_dim, num_layers)

    def forward(self, input, target_joint_names, target_heading, linear=_torch_linear):
        sel = self.chosen(input, target_joint_names, target_heading)
        return self._run(self.mlp, torch.cat([input, sel[..., None]], dim=-1).flatten(1), linear)


class EmbedTargetLocSplit(_TargetLocBase):
    """model/mdm.py:422-449 (`split`): a narrow MLP per joint, outputs concatenated along the channel axis."""

    def __init__(self, all_goal_joint_names, latent_dim, num_layers=1):
        super().__init__(all_goal_joint_names, latent_dim)
        n = len(self.extended_goal_joint_names)
        assert latent_dim % n == 0
        self.target_cond_dim, self.splited_dim = 4, latent_dim // n
        self.mini_mlps = nn.ModuleList([self._mlp(4, self.splited_dim, num_layers) for _ in range(n)])

    def forward(self, input, target_joint_names, target_heading, linear=_torch_linear):
        mi = torch.cat([input, self.chosen(input, target_joint_names, target_heading)[..., None]], dim=-1)
        return torch.cat([self._run(mlp, mi[:, j], linear) for j, mlp in enumerate(self.mini_mlps)], dim=-1)


class WeightedSum(nn.Module):
    """utils/misc.py:5-16: rows combined with learned weights normalised by their SUM (not a softmax)."""

    def __init__(self, num_rows):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(num_rows))

    def forward(self, x):          # x [..., num_rows, d]: accumulated row by row (the same bits for a sample in any batch)
        w = self.weights / self.weights.sum()
        out = w[0] * x[..., 0, :]
        for j in range(1, x.shape[-2]):
            out = out + w[j] * x[..., j, :]
        return out


class EmbedTargetLocMulti(_TargetLocBase):
    """model/mdm.py:451-479 (`multi`): an MLP per joint over its (x, y, z); the rows of the joints a sample did NOT choose are zero;
    WeightedSum over the joints.  (The reference keeps the per-joint MLPs in an nn.ParameterDict; a ModuleDict has the same keys.)"""

    def __init__(self, all_goal_joint_names, latent_dim):
        super().__init__(all_goal_joint_names, latent_dim)
        self.n_extended_goal_joints = len(self.extended_goal_joint_names)
        self.target_loc_emb = nn.ModuleDict({n: self._mlp(3, latent_dim, 1) for n in self.extended_goal_joint_names})
        self.target_all_loc_emb = WeightedSum(self.n_extended_goal_joints)

    def forward(self, input, target_joint_names, target_heading, linear=_torch_linear):
        sel = self.chosen(input, target_joint_names, target_heading)
        rows = torch.stack([self._run(self.target_loc_emb[n], input[:, j], linear) * sel[:, j:j + 1]
                            for j, n in enumerate(self.extended_goal_joint_names)], dim=1)          # [B, n_ext, d]
        return self.target_all_loc_emb(rows)


class _IdentityRot2xyz:
    """Stand-in for model/rotation2xyz.py: for data_rep='hml_vec' the callers use pose_rep='xyz', for which the
    reference returns its input unchanged (rotation2xyz.py:20-21; sample/generate.py:167)."""

    def __init__(self):
        self.smpl_model = nn.Module()

    def __call__(self, x, mask=None, pose_rep="xyz", **kw):
        if pose_rep != "xyz":
            raise NotImplementedError("SMPL forward kinematics is outside the MI355X hot path (SURVEY.md 2)")
        return x


class MDM(nn.Module):
    def __init__(self, modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1,
                 ablation=None, activation="gelu", legacy=False, data_rep='rot6d', dataset='amass', clip_dim=512,
                 arch='trans_enc', emb_trans_dec=False, clip_version=None, **kargs):
        super().__init__()
        self.legacy, self.modeltype = legacy, modeltype
        self.njoints, self.nfeats, self.num_actions = njoints, nfeats, num_actions
        self.data_rep, self.dataset = data_rep, dataset
        self.pose_rep, self.glob, self.glob_rot, self.translation = pose_rep, glob, glob_rot, translation
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.dropout, self.ablation, self.activation, self.clip_dim = dropout, ablation, activation, clip_dim
        self.action_emb = kargs.get('action_emb', None)
        self.input_feats = njoints * nfeats
        self.normalize_output = kargs.get('normalize_encoder_output', False)
        self.cond_mode = kargs.get('cond_mode', 'no_cond')
        self.cond_mask_prob = kargs.get('cond_mask_prob', 0.)
        self.mask_frames = kargs.get('mask_frames', False)
        self.arch = arch
        self.emb_policy = kargs.get('emb_policy', 'add')
        self.emb_trans_dec = emb_trans_dec
        self.pred_len = kargs.get('pred_len', 0)
        self.context_len = kargs.get('context_len', 0)
        self.total_len = self.pred_len + self.context_len
        self.is_prefix_comp = self.total_len > 0
        self.all_goal_joint_names = kargs.get('all_goal_joint_names', [])
        self.multi_target_cond = kargs.get('multi_target_cond', False)
        self.multi_encoder_type = kargs.get('multi_encoder_type', 'multi')
        self.target_enc_layers = kargs.get('target_enc_layers', 1)
        self.text_encoder_type = kargs.get('text_encoder_type', 'clip')
        self.clip_version = clip_version
        self._native_lib = kargs.get('_native_lib', None)      # tests inject the CPU emulation here
        # arithmetic of the dense contractions (include/mdm_hip.h mdm_set_precision): 'f16x3' | 'f32'.  A constructor keyword only
        # (no environment variable): a launcher that goes through the reference's factory binds it with
        # functools.partial(MDM, precision='f32')
        self.precision = kargs.get('precision', 'f16x3')
        # include/mdm_hip.h mdm_set_option values for this module's engine ({'small_gemm_max_seqs': ..., 'small_gemm_row_tiles': ...})
        self.engine_options = dict(kargs.get('engine_options', None) or {})

        if arch not in ('trans_enc', 'trans_dec'):
            raise NotImplementedError(f"arch={arch!r}: trans_enc and trans_dec (DiP) only (SURVEY.md 8f)")
        if arch == 'trans_dec' and self.total_len > kargs.get('pos_embed_max_len', 5000):
            raise ValueError(f"context_len + pred_len = {self.total_len} tokens do not fit the positional table")
        if activation != "gelu":
            raise NotImplementedError("only activation='gelu' (the reference's fixed choice, utils/model_util.py:64)")
        if data_rep == 'rot_vel' or self.emb_policy != 'add':
            raise NotImplementedError("rot_vel / emb_policy!='add' are out of scope")
        if self.cond_mode not in ('no_cond', 'text', 'action'):
            raise NotImplementedError(f"cond_mode={self.cond_mode!r}: text, action or no_cond")
        if self.multi_target_cond and self.multi_encoder_type not in ('multi', 'single', 'split'):
            raise ValueError(f"multi_encoder_type={self.multi_encoder_type!r}: multi, single or split (utils/parser_util.py:126)")
        if arch == 'trans_enc':
            if self.is_prefix_comp:
                raise NotImplementedError("prefix completion belongs to the DiP (trans_dec) path")
            if self.text_encoder_type != 'clip' and 'text' in self.cond_mode:
                raise NotImplementedError("text_encoder_type='bert' belongs to the DiP (trans_dec) path")
        else:
            if 'text' not in self.cond_mode:
                raise NotImplementedError("trans_dec: text conditioning only (model/mdm.py:261-267 builds the memory from the text)")
            if emb_trans_dec and self.is_prefix_comp:
                raise NotImplementedError("emb_trans_dec (the class-token decoder of the original paper) and prefix completion (DiP) "
                                          "are separate checkpoints upstream; the combination is not built")
            if self.text_encoder_type not in ('clip', 'bert'):
                raise ValueError('We only support [CLIP, BERT] text encoders')
            if self.text_encoder_type == 'bert':
                self.clip_dim = clip_dim = 768          # model/mdm.py:127

        self.input_process = InputProcess(data_rep, self.input_feats, latent_dim)
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout, max_len=kargs.get('pos_embed_max_len', 5000))
        if arch == 'trans_enc':
            layer = nn.TransformerEncoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size,
                                               dropout=dropout, activation=activation)
            self.seqTransEncoder = nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)
        else:
            layer = nn.TransformerDecoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size,
                                               dropout=dropout, activation=activation)
            self.seqTransDecoder = nn.TransformerDecoder(layer, num_layers=num_layers)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)
        if self.multi_target_cond:                       # model/mdm.py:67-73
            if self.multi_encoder_type == 'multi':
                self.embed_target_cond = EmbedTargetLocMulti(self.all_goal_joint_names, latent_dim)
            elif self.multi_encoder_type == 'single':
                self.embed_target_cond = EmbedTargetLocSingle(self.all_goal_joint_names, latent_dim, self.target_enc_layers)
            else:
                self.embed_target_cond = EmbedTargetLocSplit(self.all_goal_joint_names, latent_dim, self.target_enc_layers)
        if 'action' in self.cond_mode:                   # model/mdm.py:128-130
            self.embed_action = EmbedAction(self.num_actions, latent_dim)
        if 'text' in self.cond_mode:
            self.embed_text = nn.Linear(clip_dim, latent_dim)
            # the text encoder itself (CLIP / DistilBERT) is outside the hot path: callers cache y['text_embed']
            self.clip_model = self._try_load_clip(clip_version) if self.text_encoder_type == 'clip' else None
        self.output_process = OutputProcess(data_rep, self.input_feats, latent_dim, njoints, nfeats)
        self.rot2xyz = _IdentityRot2xyz()
        self._engine = None
        self._engine_key = None

    # ---- text encoder (outside the hot path: runs once per prompt batch on the host side) ------------
    @staticmethod
    def _try_load_clip(clip_version):
        try:
            import clip  # noqa: F401  (not installed in the offline image)
        except ImportError:
            return None
        model, _ = clip.load(clip_version, device='cpu', jit=False)
        model.eval()
        for p in model.parameters():
            p.requires_grad = False
        return model

    def encode_text(self, raw_text):
        """model/mdm.py:163-178 clip_encode_text (bert_encode_text :180-187 needs DistilBERT weights: cache instead)."""
        if self.text_encoder_type == 'bert':
            if getattr(self, 'clip_model', None) is None:
                raise RuntimeError("no DistilBERT attached (assign model.clip_model = load_bert(path), model/mdm.py:119): pass the "
                                   "cached embedding as y['text_embed'] = (last_hidden_state [Ntok, B, 768], pad_mask [B, Ntok])")
            enc_text, mask = self.clip_model(raw_text)        # model/mdm.py:180-187 bert_encode_text
            return enc_text.permute(1, 0, 2), ~mask
        if getattr(self, 'clip_model', None) is None:
            raise RuntimeError("CLIP is not available in this environment: pass the cached embedding as "
                               "y['text_embed'] ([1, B, clip_dim]); see sample/generate.py:130-132")
        import clip
        device = next(self.parameters()).device
        if self.dataset in ['humanml', 'kit']:
            texts = clip.tokenize(raw_text, context_length=22, truncate=True).to(device)
            texts = torch.cat([texts, torch.zeros([texts.shape[0], 77 - 22], dtype=texts.dtype, device=device)], dim=1)
        else:
            texts = clip.tokenize(raw_text, truncate=True).to(device)
        return self.clip_model.encode_text(texts).float().unsqueeze(0)

    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith('clip_model.')]

    def mask_cond(self, cond, force_mask=False):
        """model/mdm.py:153-161 (inference branches only)."""
        if force_mask:
            return torch.zeros_like(cond)
        if self.training and self.cond_mask_prob > 0.:
            raise NotImplementedError("training-time condition dropout is outside the inference hot path")
        return cond

    # ---- native engine management ---------------------------------------------------------------------
    # ---- sequence length.  Up to 224 tokens (trans_enc: 223 frames + the condition token; HumanML3D / KIT stop at 196,
    # sample/generate.py:32) a query row's attention scores stay in registers (exact softmax) and, at large batch, a sequence is one
    # GEMM tile: the kernels every BASELINE configuration runs.  Longer sequences -- the reference is bounded by its positional table
    # only (5000 rows, model/mdm.py:55, :251-253) -- run the GEMMs on row tiles at every batch size and the attention with a
    # streaming softmax over 32-key tiles (csrc/attention_long.h, round 6), both arithmetic modes.  What remains bounded: the
    # positional table itself, and frame masks WITH HOLES at 256 frames (the kernels' bitmap; prefix masks -- what collate builds --
    # have no bound).
    FAST_TOKENS = 224         # up to here: the exact-softmax attention kernels / sequence-sized GEMM tiles

    @property
    def MAX_TOKENS(self):
        return int(self.sequence_pos_encoder.pe.shape[0])

    @property
    def lead_rows(self):
        """Rows in front of the frames of a trans_dec sequence, as the library counts them (its `context_len`): the prefix frames of
        prefix completion (model/mdm.py:203-206), or ONE row for `emb_trans_dec` -- the timestep embedding as a class token
        (model/mdm.py:256-257; include/mdm_hip.h MDM_OPT_DEC_TIME_TOKEN)."""
        return 1 if self.emb_trans_dec else int(self.context_len)

    @property
    def MAX_FRAMES(self):     # trans_enc: the condition token takes one row of the positional table
        return self.MAX_TOKENS - 1 if self.arch == 'trans_enc' else self.MAX_TOKENS - self.lead_rows

    def _check_frames(self, T):
        if (T + 1 if self.arch == 'trans_enc' else self.lead_rows + T) > self.MAX_TOKENS:
            raise nat.MdmError(f"{T} frames: the sequence does not fit the positional table ({self.MAX_TOKENS} rows, "
                               f"pos_embed_max_len; model/mdm.py:55)")

    def _native_state(self):
        sd = {k: v for k, v in self.state_dict().items()
              if not k.startswith('clip_model.') and k != 'embed_timestep.sequence_pos_encoder.pe'}
        sd['sequence_pos_encoder.pe'] = sd['sequence_pos_encoder.pe'].reshape(-1, self.latent_dim)
        # the host-side condition encoders (evaluated once per call / loop by target_embedding / text_embedding below)
        sd = {k: v for k, v in sd.items() if not k.startswith(('embed_target_cond.', 'embed_action.'))}
        if 'action' in self.cond_mode:
            # emb = time_emb + mask_cond(action_embedding[y['action']]) (model/mdm.py:224-226): the class row IS the condition vector, so
            # the library's embed_text slot holds the identity (zero bias = the masked, unconditional branch) and `text_embedding`
            # hands over the gathered rows
            sd['embed_text.weight'] = torch.eye(self.latent_dim, device=sd['input_process.poseEmbedding.weight'].device)
            sd['embed_text.bias'] = torch.zeros(self.latent_dim)
        elif 'embed_text.weight' not in sd:   # no_cond: the condition token is time-only -> zero text embedding
            sd['embed_text.weight'] = torch.zeros(self.latent_dim, self.clip_dim)
            sd['embed_text.bias'] = torch.zeros(self.latent_dim)
        return sd

    @property
    def cond_dim(self):
        """Width of the per-sample condition block the library projects: clip_dim for text, latent_dim for an action row."""
        return self.latent_dim if 'action' in self.cond_mode else self.clip_dim

    def engine(self):
        """Native handle bound to the current parameters (rebuilt when they move or change).  The key is rebuilt on EVERY call
        from a fresh walk over the module's parameters -- (data_ptr, in-place version) of each of the ~150 tensors, a few
        microseconds -- so that nothing that can change a weight slips past it: in-place updates and load_state_dict bump the
        version, `load_state_dict(assign=True)`, re-assigning any nn.Parameter and `p.data = other` move the address (round 3
        cached the parameter LIST and compared its first address + the version sum only: all three of those kept serving the
        old weights, tests/test_host_logic.py::test_engine_key_sees_every_kind_of_weight_change)."""
        p = self.input_process.poseEmbedding.weight
        params = self.parameters_wo_clip()
        key = (str(p.device), self.precision, tuple(sorted({**_engine.DEFAULT_OPTIONS, **self.engine_options}.items()))) + \
            tuple((q.data_ptr(), q._version) for q in params)
        if self._engine is None or self._engine_key != key:
            cfg = dict(njoints=self.njoints, nfeats=self.nfeats, latent_dim=self.latent_dim, ff_size=self.ff_size,
                       num_layers=self.num_layers, num_heads=self.num_heads, clip_dim=self.cond_dim,
                       max_len=self.sequence_pos_encoder.pe.shape[0], mask_frames=int(bool(self.mask_frames)),
                       arch=nat.ARCH[self.arch], context_len=self.lead_rows if self.arch == 'trans_dec' else 0)
            opts = dict(self.engine_options)
            if self.arch == 'trans_dec' and self.emb_trans_dec:
                opts['dec_time_token'] = 1
            eng = Engine(cfg, lib=self._native_lib, precision=self.precision, options=opts)
            eng.bind(self._native_state(), p.device)
            self._engine, self._engine_key = eng, key
        return self._engine

    @staticmethod
    def frame_mask_lengths(valid):
        """bool [B, F] (True = the frame is a key the reference attends to, model/mdm.py:241-247) -> the int32 `lengths` array
        of the C ABI (include/mdm_hip.h): [B] valid-frame counts when every row is a PREFIX mask -- what
        data_loaders/tensors.py:3-8 builds, the kernels' fast path --, else [9 B]: counts, with -1 for the rows that have
        holes, followed by eight bitmap words per sample (bit j of word i: frame 32 i + j is valid)."""
        B, F = valid.shape
        counts = valid.sum(dim=1)
        prefix = (valid == (torch.arange(F, device=valid.device)[None, :] < counts[:, None])).all(dim=1)
        if bool(prefix.all()):
            return counts.to(torch.int32).contiguous()
        if F > 256:      # (prefix masks -- counts -- have no bound)
            raise NotImplementedError(f"frame masks WITH HOLES of more than 256 frames ({F}) do not fit the kernels' bitmap")
        bits = torch.zeros(B, 256, dtype=torch.int64, device=valid.device)
        bits[:, :F] = valid.to(torch.int64)
        words = (bits.view(B, 8, 32) << torch.arange(32, device=valid.device)).sum(dim=-1)     # < 2^32
        words = torch.where(words >= 2 ** 31, words - 2 ** 32, words)                         # two's complement int32
        counts = torch.where(prefix, counts, torch.full_like(counts, -1))
        return torch.cat([counts, words.reshape(-1)]).to(torch.int32).contiguous()

    def lengths_from_mask(self, y, T):
        """y['mask'] [B,1,1,T] bool -> int32 valid-frame counts, or None when the reference would not mask
        (model/mdm.py:241-247).  collate builds prefix masks (data_loaders/tensors.py:3-8, :22-40): those reach the
        attention kernels as valid-frame counts; any other mask as per-sample bitmaps (frame_mask_lengths).  The
        (host-synchronising) classification runs once per mask tensor: a sampling loop hands over the same `y` every step;
        the cache entry keeps the mask alive so that its address cannot be recycled."""
        mask = y.get('mask', None) if y is not None else None
        if not self.mask_frames or mask is None or mask.shape[-1] <= 1:
            return None
        key = (mask.data_ptr(), mask._version, tuple(mask.shape), str(mask.device), int(T))
        cached = getattr(self, '_len_cache', None)
        if cached is not None and cached[0] == key and cached[1] is mask:
            return cached[2]
        m = mask[..., :T].reshape(mask.shape[0], -1).to(torch.bool)
        lengths = self.frame_mask_lengths(m)
        self._len_cache = (key, mask, lengths)
        return lengths

    def target_embedding(self, y, device):
        """y['target_cond'] -> the [B, latent_dim] block the reference adds to the timestep embedding of BOTH guidance branches
        (`time_emb += mask_cond(embed_target_cond(...)[None], force_mask=y.get('target_uncond', False))`, model/mdm.py:197-199), or
        None (no target in y / force-masked).  It depends on neither x nor t: evaluated once per call or loop on the host side
        (torch, <= 8 joints x 4 numbers per sample through a SiLU MLP) and handed to the library through mdm_set_time_add; cached per
        target tensor like the frame mask."""
        if y is None or 'target_cond' not in y:
            return None
        if not self.multi_target_cond:
            raise ValueError("y['target_cond'] needs a checkpoint trained with --multi_target_cond (model/mdm.py:67-73)")
        if y.get('target_uncond', False):
            return None
        tc = y['target_cond']
        key = (tc.data_ptr(), tc._version, tuple(tc.shape), str(device), tuple(tuple(str(j) for j in n) for n in y['target_joint_names']),
               tuple(bool(h) for h in y['is_heading']), self._engine_key)
        cached = getattr(self, '_tgt_cache', None)
        if cached is not None and cached[0] == key and cached[1] is tc:
            return cached[2]
        eng = self.engine()
        with torch.no_grad():
            g = self.embed_target_cond(tc.to(device=device, dtype=torch.float32), y['target_joint_names'], y['is_heading'],
                                       linear=eng.linear)
        g = g.to(torch.float32).contiguous()
        self._tgt_cache = (key, tc, g)
        return g

    def text_embedding(self, y, device):
        """The [B, cond_dim] block the library projects with embed_text (model/mdm.py:209-218; the gathered class rows for
        cond_mode='action', :224-226, :393-396)."""
        if 'action' in self.cond_mode:
            with torch.no_grad():
                return self.embed_action(y['action'].to(device)).to(device=device, dtype=torch.float32).contiguous()
        if 'text' not in self.cond_mode:
            return None
        if 'text_embed' in y.keys():
            enc = y['text_embed']
        else:
            enc = self.encode_text(y['text'])
        if isinstance(enc, tuple):
            raise NotImplementedError("token-level (BERT) text embeddings belong to the DiP path")
        return enc.to(device=device, dtype=torch.float32).reshape(-1, self.clip_dim).contiguous()

    # ---- trans_dec (DiP) inputs -------------------------------------------------------------------------
    def _dec_inputs(self, x, y):
        """-> (prefix | None, text tokens [Ntok, B, dim], text lengths [B] int32, frame lengths [B] int32 | None)
        from the reference's `y` contract (model/mdm.py:203-216, :242-244)."""
        dev, bs = x.device, x.shape[0]
        prefix = None
        if self.emb_trans_dec:      # the class-token row: a placeholder frame the library overwrites with the timestep embedding
            prefix = torch.zeros(bs, self.njoints, self.nfeats, 1, dtype=torch.float32, device=dev)
        elif self.context_len > 0:
            prefix = y['prefix'].to(device=dev, dtype=torch.float32).contiguous()
            assert prefix.shape == (bs, self.njoints, self.nfeats, self.context_len), prefix.shape
        enc = y['text_embed'] if 'text_embed' in y.keys() else self.encode_text(y['text'])
        mask = y.get('mask', None)
        use_mask = self.mask_frames and mask is not None and mask.shape[-1] > 1
        # the derived token counts / frame counts (and the host-synchronising validity checks of the masks) are computed
        # once per (embedding, mask) pair: a sampling loop calls forward with the same y every step
        e0 = enc[0] if isinstance(enc, tuple) else enc
        key = (e0.data_ptr(), e0._version, tuple(e0.shape), str(dev), bs, x.shape[-1],
               (enc[1].data_ptr(), enc[1]._version) if isinstance(enc, tuple) else None,
               (mask.data_ptr(), mask._version) if use_mask else None)
        if (getattr(self, '_dec_cache', None) or (None,))[0] != key:
            if isinstance(enc, tuple):
                tok, pad = enc                               # [Ntok, B, 768], [B or 1, Ntok] True = no token
                pad = pad.to(dev)
                if pad.shape[0] == 1 and bs > 1:
                    pad = pad.repeat_interleave(bs, dim=0)   # single prompt for all (mdm.py:215-216)
                tl = (~pad).sum(dim=1)
                # the tokenizer pads on the right (BERT_encoder.py:28): the mask is a suffix mask <=> lengths describe it
                if not bool((pad == (torch.arange(pad.shape[1], device=dev)[None, :] >= tl[:, None])).all()):
                    raise NotImplementedError("text pad masks must be suffix masks (right-padded prompts)")
            else:                                            # CLIP: one memory token per sample (mdm.py:262)
                tok, tl = enc, torch.ones(bs, dtype=torch.int64, device=dev)
            tok = tok.to(device=dev, dtype=torch.float32).contiguous()
            if tok.dim() != 3 or tok.shape[1] != bs or tok.shape[2] != self.clip_dim or tl.shape[0] != bs or \
                    (isinstance(enc, tuple) and pad.shape[1] != tok.shape[0]):
                raise ValueError(f"y['text_embed'] must be TOKEN-major: embedding [Ntok, B={bs}, {self.clip_dim}] with pad mask "
                                 f"[B, Ntok] (what bert_encode_text returns, model/mdm.py:185); got {tuple(tok.shape)}"
                                 + (f" and {tuple(enc[1].shape)}" if isinstance(enc, tuple) else "")
                                 + " -- a [B, Ntok, dim] block (e.g. upstream's dynamic-text slice, utils/sampler_util.py:69) is "
                                   "sample-major: permute(1, 0, 2) it")
            lengths = None
            if use_mask:
                m2 = mask[..., :x.shape[-1]].reshape(bs, -1).to(dev).to(torch.bool)
                # keys of the decoder's self-attention: the context_len prefix frames (always valid) + the window's frames
                # (emb_trans_dec: the class token is never masked, model/mdm.py:245-247 -- it is the one lead row)
                lengths = self.frame_mask_lengths(torch.cat([torch.ones(bs, self.lead_rows, dtype=torch.bool, device=dev),
                                                             m2], dim=1))
            # the entry keeps the SOURCE tensors alive: a later batch of the same shape must not be able to land on a
            # recycled address with _version 0 and hit this entry
            self._dec_cache = (key, (tok, tl.to(torch.int32).contiguous(), lengths), (enc, mask))
        tok, tl, lengths = self._dec_cache[1]
        return prefix, tok, tl, lengths

    def _forward_dec(self, x, timesteps, y, branches):
        x = x.to(torch.float32).contiguous()
        assert x.shape[1] == self.njoints and x.shape[2] == self.nfeats
        ts = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        prefix, enc, tl, lengths = self._dec_inputs(x, y)
        eng = self.engine()
        return eng.forward_dec(x, prefix, ts, enc, tl, lengths, branches, time_add=self.target_embedding(y, x.device))

    # ---- the seam ----------------------------------------------------------------------------------
    def forward(self, x, timesteps, y=None):
        """x: [bs, njoints, nfeats, nframes]; timesteps: [bs] int; y: dict (model/mdm.py:189-194)."""
        if self.training:
            raise NotImplementedError("inference only: call .eval() (sample/generate.py:96)")
        y = {} if y is None else y
        if self.arch == 'trans_dec':
            uncond = bool(y.get('uncond', False))
            return self._forward_dec(x, timesteps, y, nat.BRANCH_UNCOND if uncond else nat.BRANCH_COND)
        if 'prefix' in y:
            raise NotImplementedError("y['prefix'] (prefix completion) belongs to the trans_dec (DiP) path")
        bs, njoints, nfeats, nframes = x.shape
        assert njoints == self.njoints and nfeats == self.nfeats
        self._check_frames(nframes)
        eng = self.engine()
        x = x.to(torch.float32).contiguous()
        ts = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        uncond = bool(y.get('uncond', False)) or self.cond_mode == 'no_cond'
        te = None if uncond else self.text_embedding(y, x.device)
        if te is not None and te.shape[0] != bs:
            raise ValueError(f"text_embed batch {te.shape[0]} != x batch {bs}")
        lengths = self.lengths_from_mask(y, nframes)
        if lengths is not None:
            lengths = lengths.to(x.device)
        return eng.forward(x, ts, te, lengths, nat.BRANCH_UNCOND if uncond else nat.BRANCH_COND,
                           time_add=self.target_embedding(y, x.device))

    def forward_both(self, x, timesteps, y):
        """cond and uncond branches batched through one native call -> (out_cond, out_uncond)."""
        bs = x.shape[0]
        if self.arch == 'trans_dec':
            out = self._forward_dec(x, timesteps, y, nat.BRANCH_BOTH)
            return out[:bs], out[bs:]
        self._check_frames(x.shape[-1])
        eng = self.engine()
        x = x.to(torch.float32).contiguous()
        ts = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        te = self.text_embedding(y, x.device)
        lengths = self.lengths_from_mask(y, x.shape[-1])
        if lengths is not None:
            lengths = lengths.to(x.device)
        out = eng.forward(x, ts, te, lengths, nat.BRANCH_BOTH, time_add=self.target_embedding(y, x.device))
        return out[:bs], out[bs:]

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._engine = None   # parameters moved: rebind lazily
        self._len_cache = self._dec_cache = self._tgt_cache = None
        return r
