"""FRESCOAttnProcessor2_0: the diffusers AttnProcessor the reference installs on every attention of
`up_blocks.2*` / `up_blocks.3*` (src/diffusion_hacked.py:142-403), with the three attention passes
running in libfresco_hip.so.

Call protocol, constructor signature, the attributes read from the diffusers `Attention` module and
the controller interaction are the reference's; the data path is not:
  * head split / merge are index arithmetic inside the kernels (no (B,H,L,D) copies);
  * efficient cross-frame K/V selection + repeat (225-247) is one packed K / V^T image per CFG half,
    shared by all frames;
  * spatial-guided pass (257-288): no HW x HW eye mask is materialised; `key * 0.2` is a logit scale;
  * temporal pass (309-367): one fused gather / N x N masked softmax / scatter kernel;
  * `attn.to_q/to_k/to_v` run as one fresco_linear launch that reads the hidden states once (and
    `to_out[0]` at C = 320) when they are plain nn.Linear modules; wrapped modules keep their forward.
"""
import math
import warnings
import weakref

import torch

from . import _lib, ops
from .control import AttentionControl


def _plain_linear(m, with_bias):
    """an unwrapped nn.Linear whose forward is exactly x W^T (+ b): safe to run through fresco_linear"""
    return (type(m) is torch.nn.Linear and (m.bias is not None) == with_bias and m.weight.is_cuda
            and ops.linear_supported(m.in_features, m.out_features, m.weight.dtype))


class FRESCOAttnProcessor2_0:
    def __init__(self, unet_chunk_size=2, controller=None):
        _lib.load()  # fail here, loudly, when the HIP library is absent
        self.unet_chunk_size = unet_chunk_size
        self.controller = controller
        self._ws = ops.Workspace()
        self._rows_cache = {}
        self.shard = None  # fresco_amd.dist.FrameShard for frame-parallel multi-GPU runs
        self.fuse_projections = True  # q/k/v (and to_out at C = 320) through fresco_linear when they are plain Linears
        # cross-frame-only calls (no temporal pass) read K and V of the selected tokens only: project just those
        self.sparse_kv_projection = True
        # ... and (round 6) project them INSIDE the key pack: one launch instead of fresco_linear_rows + kv_pack, K and V
        # never reach HBM (fresco_attn_fwd_kvproj; plain bias-free fp16 to_k / to_v of a supported width only)
        self.fuse_kv_pack = True

    # ---- fused projections ---------------------------------------------------------------------------
    def _project(self, attn, x, names, outs=None, x_rows=None):
        """[attn.<name>(x) for name in names] in ONE launch that reads x once (fresco_linear), when every module
        is a plain bias-free fp16 nn.Linear of a supported width; otherwise the modules are called as the
        reference calls them (wrapped / LoRA / quantised layers keep their own forward).  The kernel reads each
        module's weight where it lives: nothing is stacked or cached, in-place weight updates are always seen."""
        mods = [getattr(attn, n) for n in names]
        if (self.fuse_projections and x.dtype == torch.float16 and x.is_cuda
                and all(_plain_linear(m, False) for m in mods)
                and len({(m.in_features, m.out_features) for m in mods}) == 1
                and all(m.weight.is_contiguous() for m in mods)):
            # (x_rows come from _sel_rows: positions of a mask's True entries, in range by construction)
            return ops.linear(x, [m.weight.detach() for m in mods], None, outs, x_rows=x_rows, x_rows_trusted=True)
        if x_rows is not None:  # (modules that keep their own forward: gather first)
            x = x.reshape(-1, x.shape[-1]).index_select(0, x_rows.long())
        res = [m(x) for m in mods]
        if outs is not None:
            for o, r in zip(outs, res):
                o.copy_(r)
            return outs
        return res

    def _project_out(self, attn, hs):
        lin = attn.to_out[0]
        # (round 3 left the C = 640 case to the library GEMM; on repeated measurement the two tie within the box-to-box
        # spread -- 27-30 us here, 24-33 us hipBLASLt at (16384, 640, 640) -- so the hot path now has no library GEMM)
        if (self.fuse_projections and hs.dtype == torch.float16 and hs.is_cuda and _plain_linear(lin, True)
                and ops.linear_supported(lin.in_features, lin.out_features, hs.dtype)):
            return ops.linear(hs, [lin.weight.detach()], [lin.bias.detach()])[0]
        return lin(hs)

    # flat int32 indices of the True entries of a (N, HW) mask, cached per mask tensor
    def _kv_rows(self, mask, as_long=False):
        key = (mask.data_ptr(), tuple(mask.shape), mask._version)
        hit = self._rows_cache.get(key)
        if hit is None or hit[0]() is not mask:  # the weakref guards against a recycled address
            if len(self._rows_cache) > 16:
                self._rows_cache.clear()
            rows64 = torch.nonzero(mask.reshape(-1), as_tuple=False).squeeze(1).contiguous()
            hit = (weakref.ref(mask), rows64.to(torch.int32), rows64)
            self._rows_cache[key] = hit
        return hit[2] if as_long else hit[1]

    def _sel_rows(self, mask, chunk, n_frames, hw, device):
        """flat int32 row indices (into the (chunk * n_frames * hw, C) hidden states) of the cross-frame keys of every
        CFG half, in the order the pass enumerates them; cached per mask (None: frame 0 of each half)"""
        key = ("sel", None if mask is None else (mask.data_ptr(), tuple(mask.shape), mask._version), chunk, n_frames, hw)
        hit = self._rows_cache.get(key)
        if hit is None or (mask is not None and hit[0]() is not mask):
            base = (torch.arange(hw, device=device) if mask is None else self._kv_rows(mask, as_long=True).to(device))
            rows = torch.cat([base + c * n_frames * hw for c in range(chunk)]).to(torch.int32).contiguous()
            # the table is handed to fresco_linear_rows unchecked on every later call: check it once, here
            if rows.numel() and not (0 <= int(rows.min()) and int(rows.max()) < chunk * n_frames * hw):
                raise ValueError("fresco_amd: cross-frame mask of shape %s addresses tokens outside the (%d, %d, %d) batch"
                                 % (tuple(mask.shape), chunk, n_frames, hw))
            hit = (weakref.ref(mask) if mask is not None else None, rows)
            self._rows_cache[key] = hit
        return hit[1]

    def _warn_rounding(self, dtype):
        """fp32 / bf16 pipelines: said once per processor (values beyond +-65504 would become inf in the fp16 kernels)"""
        if not getattr(self, "_warned_rounding", False):
            self._warned_rounding = True
            warnings.warn("fresco_amd: %s activations are rounded to fp16 for the attention kernels and the result is "
                          "cast back (the reference computes in the input dtype); |q|, |k|, |v| must stay below 65504"
                          % dtype, RuntimeWarning, stacklevel=3)

    # ---- attention_mask (reference :192-196, 303-305) ---------------------------------------------------------------
    # The pipeline never passes one to these layers (src/pipe_FRESCO.py:201-209), so this is a plain-speed side path with NO
    # kernel of its own: an additive per-(batch, head, key) bias b is an extra contraction dimension --
    # [q, 1/scale] . [k, b] * scale = q.k * scale + b -- so q, k, v are padded to the next head dim the flash kernel
    # supports, column D carries 1/scale resp. the bias, and the ordinary kernel runs.
    _HEAD_DIMS = (8, 16, 32, 40, 64, 80, 96, 128)

    def _mask_bias(self, attn, attention_mask, seq_len, batch_size):
        """-> additive bias (B, heads, Lk) fp32 from the mask forms the reference accepts: whatever
        `attn.prepare_attention_mask` returns ((B*heads, 1, Lk), reference :193), or a (B, Lk) / (B, 1, Lk) tensor;
        bool masks mean "attend where True".  Query-dependent masks are not supported."""
        m = attention_mask
        if hasattr(attn, "prepare_attention_mask"):
            m = attn.prepare_attention_mask(m, seq_len, batch_size)
            m = m.view(batch_size, attn.heads, -1, m.shape[-1])
        else:
            if m.dim() == 2:
                m = m[:, None, None, :]
            elif m.dim() == 3:
                m = m[:, None, :, :]
            m = m.expand(batch_size, attn.heads, m.shape[-2], m.shape[-1])
        if m.shape[2] != 1:
            raise NotImplementedError("fresco_amd: query-dependent attention masks are not supported (mask shape %s)"
                                      % (tuple(attention_mask.shape),))
        m = m[:, :, 0, :]
        if m.dtype == torch.bool:
            m = torch.zeros(m.shape, dtype=torch.float32, device=m.device).masked_fill_(~m, -6.0e4)
        # (-6e4: "never attended" inside fp16 range; the reference's own UNet builds -10000, diffusion_hacked.py:569)
        return m.float().clamp_min(-6.0e4).contiguous()

    def _masked_attention(self, q, k, v, heads, scale, bias):
        B, Lq, C = q.shape
        Lk = k.shape[1]
        D = C // heads
        if bias.shape != (B, heads, Lk) or k.shape[0] != B:
            raise ValueError("fresco_amd: attention_mask of shape %s does not match %d batches x %d heads x %d keys"
                             % (tuple(bias.shape), B, heads, Lk))
        Dp = next((d for d in self._HEAD_DIMS if d > D), None)
        if Dp is None:
            raise NotImplementedError("fresco_amd: attention_mask with head dim %d (no larger kernel head dim)" % D)

        def pad(x, L, col):
            xp = x.new_zeros(B, L, heads, Dp)
            xp[..., :D] = x.view(B, L, heads, D)
            if col is not None:
                xp[..., D] = col
            return xp.view(B, L, heads * Dp)

        # the bias rides in the contraction as bias / scale: keep that inside the kernel's +-6e4 running-max window on every
        # head-dim instantiation (a fully masked row then gives the uniform softmax of the reference's -10000 masks, not
        # 0 / 0); exp(-5e4 * scale) is still exactly 0 next to any unmasked key
        bias = bias.clamp_min(-5.0e4 * scale)
        qp = pad(q, Lq, 1.0 / scale)
        kp = pad(k, Lk, bias.transpose(1, 2).to(k.dtype))
        vp = pad(v, Lk, None)
        out = ops.attention(qp, kp, vp, heads, scale, workspace=self._ws)
        return out.view(B, Lq, heads, Dp)[..., :D].reshape(B, Lq, C)

    def _cf_mask(self, ctrl, hw):
        """the cross-frame key mask of this feature scale (None: every frame attends to frame 0 only)"""
        mask = None
        if ctrl.attn_mask is not None:
            for m in ctrl.attn_mask:
                if m.shape[1] == hw:
                    mask = m
        return mask

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        residual = hidden_states
        if attn.spatial_norm is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            batch_size, channel, height, width = hidden_states.shape
            hidden_states = hidden_states.view(batch_size, channel, height * width).transpose(1, 2)
        batch_size = hidden_states.shape[0]
        mask_bias = None
        if attention_mask is not None:  # reference :192-196
            seq_len = hidden_states.shape[1] if encoder_hidden_states is None else encoder_hidden_states.shape[1]
            mask_bias = self._mask_bias(attn, attention_mask, seq_len, batch_size)
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)

        ctrl = self.controller
        crossattn = encoder_hidden_states is not None
        fused_kv = None
        if not crossattn:
            encoder_hidden_states = hidden_states
            if ctrl and ctrl.store:
                ctrl(hidden_states.detach().clone())
            if self.shard is not None and ctrl and (ctrl.use_cfattn or ctrl.use_interattn):
                if mask_bias is not None:
                    raise ValueError("fresco_amd: attention_mask is not supported in the frame-sharded FRESCO branch")
                return self._sharded_self_attention(attn, hidden_states, residual, input_ndim)
            sparse_kv = (self.sparse_kv_projection and bool(ctrl) and ctrl.use_cfattn and not ctrl.use_interattn
                         and mask_bias is None and hidden_states.shape[0] % self.unet_chunk_size == 0)
            if sparse_kv:
                # (the row table below is built from the mask's own (frame, pixel) grid and handed to the kernel
                # unchecked: a mask of another batch size takes the general path, whose row table IS bounds-checked)
                m_ = self._cf_mask(ctrl, hidden_states.shape[1])
                sparse_kv = m_ is None or (m_.dim() == 2 and
                                           m_.shape[0] == hidden_states.shape[0] // self.unet_chunk_size)
            if sparse_kv:
                # the only reader of K and V is the cross-frame pass, which gathers frame 0 and the selected tokens of
                # the other frames (225-247): project exactly those rows, in the order the pass enumerates them
                (query,) = self._project(attn, hidden_states, ("to_q",))
                chunk_, hw_ = self.unet_chunk_size, hidden_states.shape[1]
                mask = self._cf_mask(ctrl, hw_)
                nf = hidden_states.shape[0] // chunk_
                rows_all = self._sel_rows(mask, chunk_, nf, hw_, hidden_states.device)  # flat rows of both CFG halves
                if (self.fuse_kv_pack and self.fuse_projections and hidden_states.dtype == torch.float16
                        and hidden_states.is_cuda and not ctrl.use_intraattn
                        and _plain_linear(attn.to_k, False) and _plain_linear(attn.to_v, False)
                        and attn.to_k.weight.is_contiguous() and attn.to_v.weight.is_contiguous()
                        and attn.to_k.out_features == attn.to_v.out_features == query.shape[-1]
                        and attn.to_k.out_features % attn.heads == 0
                        and ops.attention_kvproj_supported(attn.heads, attn.to_k.out_features // attn.heads,
                                                           attn.to_k.in_features)):
                    # K | V of the selected rows are projected inside the key pack of the cross-frame pass below
                    fused_kv = (rows_all, attn.to_k.weight.detach(), attn.to_v.weight.detach(), rows_all.numel() // chunk_)
                    key = value = None
                else:
                    key, value = self._project(attn, hidden_states, ("to_k", "to_v"), x_rows=rows_all)
                    key, value = key.view(chunk_, -1, key.shape[-1]), value.view(chunk_, -1, value.shape[-1])
            else:
                query, key, value = self._project(attn, hidden_states, ("to_q", "to_k", "to_v"))
        else:
            sparse_kv = False
            query = attn.to_q(hidden_states)
            if attn.norm_cross:
                encoder_hidden_states = attn.norm_encoder_hidden_states(encoder_hidden_states)
            key = attn.to_k(encoder_hidden_states)
            value = attn.to_v(encoder_hidden_states)

        # the kernels compute in fp16 (the dtype the pipeline runs its UNet in); fp32 / bf16 activations are rounded
        # to fp16 after the projections and the result is cast back: same HIP path, no eager branch
        out_dtype = query.dtype
        if out_dtype != torch.float16:
            self._warn_rounding(out_dtype)
            query, key, value = query.half(), key.half(), value.half()

        heads = attn.heads
        head_dim = query.shape[-1] // heads
        sm_scale = 1.0 / math.sqrt(head_dim)
        fresco = bool(ctrl) and not crossattn
        chunk = self.unet_chunk_size

        # spatial-guided pass: the current query becomes the VALUE of an attention over the stored
        # features of the input video (diffusion_hacked.py:257-288)
        q_att = query
        if fresco and ctrl.use_intraattn:
            ref = ctrl(None)
            assert ref.shape == encoder_hidden_states.shape
            q_ref, k_ref = self._project(attn, ref, ("to_q", "to_k"))
            if q_ref.dtype != torch.float16:
                q_ref, k_ref = q_ref.half(), k_ref.half()
            q_att = ops.attention(q_ref, k_ref, query, heads, ctrl.intraattn_scale_factor * sm_scale,
                                  diag_bias=float(ctrl.intraattn_bias), workspace=self._ws)

        # main pass: efficient cross-frame attention (225-247, 303-305) or plain attention
        if mask_bias is not None:
            if fresco and ctrl.use_cfattn:
                if self._cf_mask(ctrl, key.shape[1]) is not None or sparse_kv:
                    # with a controller.attn_mask of this scale the reference hands SDPA a mask shaped for
                    # `sequence_length` keys next to M != sequence_length cross-frame keys (:239-247, 303-305) and cannot
                    # run either
                    raise ValueError("fresco_amd: attention_mask cannot be combined with the masked cross-frame key set "
                                     "(the mask addresses %d keys, the cross-frame pass has another key set)" % mask_bias.shape[-1])
                # controller.attn_mask None (or no mask of this scale): every frame attends to frame 0's HW keys
                # (former_frame_index = [0] * N, :227, 237, 244) and the mask addresses exactly those: the reference's
                # SDPA accepts it, so do we (a combination the pipeline never produces, src/pipe_FRESCO.py:201-209)
                nf = key.shape[0] // chunk
                k0 = key.view(chunk, nf, key.shape[1], -1)[:, :1].expand(-1, nf, -1, -1).reshape(key.shape)
                v0 = value.view(chunk, nf, value.shape[1], -1)[:, :1].expand(-1, nf, -1, -1).reshape(value.shape)
                hs = self._masked_attention(q_att, k0, v0, heads, sm_scale, mask_bias)
            else:
                hs = self._masked_attention(q_att, key, value, heads, sm_scale, mask_bias)
        elif fresco and ctrl.use_cfattn and sparse_kv and fused_kv is not None:
            hs = ops.attention_kvproj(q_att, hidden_states, fused_kv[0], fused_kv[1], fused_kv[2], heads, sm_scale,
                                      n_groups=chunk, M=fused_kv[3], workspace=self._ws)
        elif fresco and ctrl.use_cfattn and sparse_kv:
            hs = ops.attention(q_att, key, value, heads, sm_scale, n_groups=chunk, M=key.shape[1],
                               group_rows=key.shape[1], workspace=self._ws)
        elif fresco and ctrl.use_cfattn:
            video_length = key.shape[0] // chunk
            hw = key.shape[1]
            mask = self._cf_mask(ctrl, hw)
            rows = self._kv_rows(mask) if mask is not None else None
            hs = ops.attention(q_att, key, value, heads, sm_scale, kv_rows=rows, n_groups=chunk,
                               M=hw if rows is None else rows.numel(), group_rows=video_length * hw,
                               workspace=self._ws)
        else:
            hs = ops.attention(q_att, key, value, heads, sm_scale, workspace=self._ws)

        # temporal-guided pass along the flow trajectories (309-367)
        if fresco and ctrl.use_interattn:
            fwd_mapping = interattn_mask = None
            paras = ctrl.interattn_paras
            for i, f in enumerate(paras["fwd_mappings"]):
                if f.shape[2] == hs.shape[1]:
                    fwd_mapping = f
                    interattn_mask = paras["interattn_masks"][i]
            if fwd_mapping is None:
                raise ValueError("fresco_amd: no temporal-attention parameters for %d tokens" % hs.shape[1])
            hs = ops.temporal_attention(query, key, hs, fwd_mapping, interattn_mask, heads,
                                        ctrl.interattn_scale_factor * sm_scale, chunk)

        hs = hs.to(out_dtype)
        hs = self._project_out(attn, hs)
        hs = attn.to_out[1](hs)
        if input_ndim == 4:
            hs = hs.transpose(-1, -2).reshape(batch_size, channel, height, width)
        if attn.residual_connection:
            hs = hs + residual
        if attn.rescale_output_factor != 1.0:  # x / 1.0 == x exactly: skip the elementwise pass
            hs = hs / attn.rescale_output_factor
        return hs


def _sharded_self_attention(self, attn, hidden_states, residual, input_ndim):
    """Frame-parallel form of the FRESCO self-attention branch (fresco_amd/dist.py): this rank holds
    `shard.n_loc` frames of both CFG halves.  Cross-frame keys: broadcast of frame 0 + all-gather of the
    other frames' selected rows; temporal pass: all-to-all to trajectory shards and back; the rest is local."""
    if input_ndim != 3:
        raise NotImplementedError("fresco_amd: frame-sharded attention expects (B, HW, C) hidden states")
    out_dtype = hidden_states.dtype
    if out_dtype != torch.float16:  # same policy as the single-GPU path: modules in their dtype, kernels in fp16
        self._warn_rounding(out_dtype)
    ctrl, sh = self.controller, self.shard
    chunk = self.unet_chunk_size
    heads = attn.heads
    B_loc, hw, _ = hidden_states.shape
    C = getattr(attn.to_k, "out_features", hidden_states.shape[-1])  # inner dim (= hidden width in SD-1.5 attn1)
    head_dim = C // heads
    sm_scale = 1.0 / math.sqrt(head_dim)
    assert B_loc == sh.B_loc and chunk == sh.chunk
    # q, k, v in one pass over the hidden states; K and V land fused per row (K | V), the layout of the exchange
    query = torch.empty(B_loc, hw, C, dtype=torch.float16, device=hidden_states.device)
    kv_loc = torch.empty(B_loc, hw, 2 * C, dtype=torch.float16, device=hidden_states.device)
    key, value = kv_loc[..., :C], kv_loc[..., C:]
    self._project(attn, hidden_states, ("to_q", "to_k", "to_v"), outs=[query, key, value])
    works = []
    if ctrl.use_cfattn:
        mask = None
        if ctrl.attn_mask is not None:
            for m in ctrl.attn_mask:
                if m.shape[1] == hw:
                    mask = m
        plan = sh.cf_plan(mask, hw, key.device)
        # launched before the local work they overlap with
        kvbuf, works = sh.exchange_cf(kv_loc, plan)
    q_att = query
    if ctrl.use_intraattn:
        ref = ctrl(None)
        assert ref.shape == hidden_states.shape
        q_ref, k_ref = self._project(attn, ref, ("to_q", "to_k"))
        if q_ref.dtype != torch.float16:
            q_ref, k_ref = q_ref.half(), k_ref.half()
        q_att = ops.attention(q_ref, k_ref, query, heads,
                              ctrl.intraattn_scale_factor * sm_scale, diag_bias=float(ctrl.intraattn_bias),
                              workspace=self._ws)
    for w in works:
        w.wait()
    if ctrl.use_cfattn:
        flat = kvbuf.view(-1, 2 * C)
        hs = ops.attention(q_att, flat[:, :C], flat[:, C:], heads, sm_scale, kv_rows=plan["kv_table"],
                           n_groups=chunk, M=plan["M"], group_rows=plan["kv_group_rows"], workspace=self._ws)
    else:
        hs = ops.attention(q_att, key, value, heads, sm_scale, workspace=self._ws)
    if ctrl.use_interattn:
        fwd_mapping = interattn_mask = None
        paras = ctrl.interattn_paras
        for i, f in enumerate(paras["fwd_mappings"]):
            if f.shape[2] == hw:
                fwd_mapping = f
                interattn_mask = paras["interattn_masks"][i]
        if fwd_mapping is None:
            raise ValueError("fresco_amd: no temporal-attention parameters for %d tokens" % hw)
        hs = sh.temporal(query, key, hs, fwd_mapping, interattn_mask, heads,
                         ctrl.interattn_scale_factor * sm_scale)
    hs = self._project_out(attn, hs.to(out_dtype))
    hs = attn.to_out[1](hs)
    if attn.residual_connection:
        hs = hs + residual
    if attn.rescale_output_factor != 1.0:
        hs = hs / attn.rescale_output_factor
    return hs


FRESCOAttnProcessor2_0._sharded_self_attention = _sharded_self_attention


def apply_FRESCO_attn(pipe):
    """Install one shared FRESCO processor on the decoder attentions (diffusion_hacked.py:390-403).
    The other attentions keep diffusers' stock AttnProcessor2_0.

    Narrower than the reference on purpose (there is no eager fallback behind the HIP kernels): the processor takes
    CUDA hidden states and computes the attention in fp16 (the dtype run_fresco.py runs the UNet in, :63-80): fp32 /
    bf16 activations are rounded to fp16 after the projections and the result is cast back (one RuntimeWarning per
    processor; frame-sharded runs follow the same policy); an `attention_mask` (the pipeline never passes one to these layers)
    is honoured on the plain / cross-attention path through a padded-head-dim side path, and rejected where the reference
    itself cannot use it (together with cross-frame attention) or where it depends on the query."""
    from diffusers.models.attention_processor import AttnProcessor2_0

    frescoProc = FRESCOAttnProcessor2_0(2, AttentionControl())
    attnProc = AttnProcessor2_0()
    procs = {}
    for k in pipe.unet.attn_processors.keys():
        procs[k] = frescoProc if (k.startswith("up_blocks.2") or k.startswith("up_blocks.3")) else attnProc
    pipe.unet.set_attn_processor(procs)
    return frescoProc
