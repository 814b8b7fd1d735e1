// loops.h -- part of the ONE translation unit csrc/mdm_api.hip (the C ABI of libmdm_hip.so); split out of it in round 6
// (source health, no behaviour change).  The sampling loops as single native calls: mdm_sample_loop (gaussian_diffusion.py:591-727, :876-990) and
// mdm_sample_loop_dec (one DiP prediction window).
#pragma once
// (included inside mdm_api.hip's extern "C" block: these ARE exported entry points of include/mdm_hip.h)

int mdm_sample_loop(mdm_model_t* m, const mdm_sample_params_t* p, float* x, void* ws_dev, size_t ws_bytes,
                    void* stream) {
  ChainGuard chain_guard(stream);
  if (int rc = check_ready(m)) return rc;
  if (m->cfg.arch != MDM_ARCH_TRANS_ENC) return fail(MDM_ESTATE, "mdm_sample_loop: the fused loop drives the trans_enc denoiser");
  if (p == nullptr || x == nullptr || ws_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop: null pointer");
  const int B = p->B, T = p->T;
  TimeAddScope time_add(m, B, "mdm_sample_loop");
  if (time_add.rc) return time_add.rc;
  if (B <= 0 || T <= 0 || T + 1 > m->cfg.max_len) return fail(MDM_EINVAL, "mdm_sample_loop: need B >= 1 and 1 <= T < the positional table's length");
  if (p->num_timesteps <= 0 || p->start_index < 0 || p->start_index >= p->num_timesteps)
    return fail(MDM_EINVAL, "mdm_sample_loop: bad start_index / num_timesteps");
  if (!p->a_x0 || !p->a_xt || !p->sigma || !p->timestep_map) return fail(MDM_EINVAL, "mdm_sample_loop: null schedule table");
  if ((p->inpaint_mask_dev == nullptr) != (p->inpaint_motion_dev == nullptr))
    return fail(MDM_EINVAL, "mdm_sample_loop: inpainting needs mask and motion");
  const bool cfg = p->scale_dev != nullptr;
  const bool uncond_only = !cfg && (p->force_uncond || p->text_embed_dev == nullptr);
  if (cfg && p->text_embed_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop: CFG needs text_embed");
  if (p->num_dump > 0 && (p->dump_steps == nullptr || p->dump_dev == nullptr)) return fail(MDM_EINVAL, "mdm_sample_loop: dump buffers missing");
  for (int i = 0; i <= p->start_index; ++i)
    if (p->timestep_map[i] < 0 || p->timestep_map[i] >= m->cfg.max_len) return fail(MDM_EINVAL, "mdm_sample_loop: timestep outside the positional table");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nbranch = cfg ? 2 : 1, nseq = nbranch * B, S = T + 1, D = m->cfg.latent_dim;
  Workspace ws = carve(m, nseq, T, ws_dev);
  if (ws_bytes < ws.bytes) return fail(MDM_ENOSPC, "mdm_sample_loop: workspace too small");
  const int* len = m->cfg.mask_frames ? p->lengths_dev : nullptr;
  const size_t per_sample = (size_t)m->jf * T;

  // step-invariant: embed_text(cond) once per loop (gaussian_diffusion.py:633-635 caches the encoder side;
  // the Linear on top is also constant across steps)
  if (!uncond_only)
    if (int rc = launch_linear(nullptr, p->text_embed_dev, m->cfg.clip_dim, m->W("embed_text.weight"), m->W("embed_text.bias"),
                               nullptr, ws.cond, B, D, m->cfg.clip_dim, ACT_NONE, 0, 1.f, s)) return rc;
  const int uncond_from = uncond_only ? 0 : 1;

  int dump_i = 0, k = 0;
  for (int i = p->start_index; i >= 0; --i, ++k) {
    // frame tokens + condition token for model timestep timestep_map[i]
    {
      PoseGatherLoader al{x, T, m->jf, B * T};
      RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
      const bool x3 = m->precision == MDM_PREC_F16X3;
      EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, T, S, D, nbranch,
                       x3 ? ws.tokh : nullptr, x3 ? ws.tokl : nullptr};
      if (use_embed_x3(m, T)) {   // (token 0 of every sequence rides in the transpose kernel of the frame embedding)
        const CondTokArgs ct{ws.tok, ws.cond, m->W("embed_text.bias"), m->time_table, nullptr, (int)p->timestep_map[i],
                             m->W("sequence_pos_encoder.pe"), B, S, D, uncond_from, (int)m->cfg.max_len, ws.tokh, ws.tokl, m->time_add};
        if (int rc = embed_frames_x3(m, ws, x, B, T, nbranch, s, ct)) return rc;
      } else {
        {
          ProfScope ps(&m->prof, MDM_PROF_EMBED, 2.0 * B * T * (double)D * m->jf, s);
          launch_gemm_f32(al, bl, ep, B * T, D, m->jf_pad, s);
        }
        if (int rc = rt_launch_status()) return rc;
        ProfScope ps(&m->prof, MDM_PROF_ELEMENTWISE, 0.0, s);
        MDM_LAUNCH(cond_token_kernel, dim3(nseq), dim3(128), 0, s, ws.tok, (const float*)ws.cond,
                   m->W("embed_text.bias"), (const float*)m->time_table, (const long long*)nullptr,
                   (int)p->timestep_map[i], m->W("sequence_pos_encoder.pe"), B, S, D, uncond_from,
                   (int)m->cfg.max_len, x3 ? ws.tokh : (p16_t*)nullptr, x3 ? ws.tokl : (p16_t*)nullptr, m->time_add);
        if (int rc = rt_launch_status()) return rc;
      }
    }
    if (int rc = encoder(m, ws, nseq, B, S, len, s)) return rc;
    // this step's eps: injected, or the counter-based stream -- drawn inline by the split-precision tail kernel, into the
    // (now dead) attention buffer for the exact-fp32 OutputProcess epilogue
    const bool x3mode = m->precision == MDM_PREC_F16X3;
    const float* step_noise = nullptr;
    if (p->sigma[i] != 0.f) {
      if (p->noise_dev != nullptr) step_noise = p->noise_dev + (size_t)k * B * per_sample;
      else if (!x3mode) {
        ProfScope ps(&m->prof, MDM_PROF_ELEMENTWISE, 0.0, s);
        if (int rc = launch_randn(ws.att, nullptr, nullptr, 0.f, 1.f, B, (int)per_sample, p->seed, p->sample_base,
                                  (uint32_t)(1 + k), (uint32_t)(p->const_noise != 0), stream)) return rc;
        step_noise = ws.att;
      }
    }
    // OutputProcess + CFG combine + sampler update, in place on x
    if (x3mode) {
      if (int rc = outproj_x3(m, ws, nseq, B, T, cfg ? p->scale_dev : nullptr, 1, x, (i == 0) ? p->x0_dev : nullptr, x,
                              NoiseSource{step_noise, p->seed, p->sample_base, (uint32_t)(1 + k), (uint32_t)(p->const_noise != 0)},
                              p->inpaint_mask_dev,
                              p->inpaint_motion_dev,
                              StepCoefs{p->a_x0[i], p->a_xt[i], p->sigma[i], p->clip_denoised}, s)) return rc;
    } else {
      RowMajorLoader al{m->W("output_process.poseFinal.weight"), D, m->jf, D};
      CfgTokenLoader bl{ws.tok, cfg ? p->scale_dev : nullptr, B, T, S, D, B * T};
      OutProjEpilogue ep{};
      ep.bias = m->W("output_process.poseFinal.bias");
      ep.out = x;
      ep.x0_out = (i == 0) ? p->x0_dev : nullptr;
      ep.x_t = x;
      ep.inpaint_mask = p->inpaint_mask_dev;
      ep.inpaint_motion = p->inpaint_motion_dev;
      ep.T = T; ep.JF = m->jf; ep.mode = 1;
      ep.co = StepCoefs{p->a_x0[i], p->a_xt[i], p->sigma[i], p->clip_denoised};
      ep.noise = step_noise;
      ProfScope ps(&m->prof, MDM_PROF_OUTPROJ, 2.0 * B * T * (double)D * m->jf, s);
      launch_gemm_f32(al, bl, ep, m->jf, B * T, D, s);
      if (int rc = rt_launch_status()) return rc;
    }
    if (dump_i < p->num_dump && p->dump_steps[dump_i] == k) {
      if (int rc = rt_copy(p->dump_dev + (size_t)dump_i * B * per_sample, x, (size_t)B * per_sample * sizeof(float), s)) return rc;
      ++dump_i;
    }
  }
  return MDM_OK;
}

int mdm_sample_loop_dec(mdm_model_t* m, const mdm_sample_dec_params_t* pd, float* x, void* ws_dev, size_t ws_bytes,
                        void* stream) {
  ChainGuard chain_guard(stream);
  if (int rc = check_ready(m)) return rc;
  if (pd == nullptr || x == nullptr || ws_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop_dec: null pointer");
  const mdm_sample_params_t* p = &pd->loop;
  const int B = p->B, P = p->T, ntok = pd->ntok;
  TimeAddScope time_add(m, B, "mdm_sample_loop_dec");
  if (time_add.rc) return time_add.rc;
  if (int rc = check_dec_shapes(m, "mdm_sample_loop_dec", pd->prefix_dev, B, P, ntok)) return rc;
  if (pd->text_lengths_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop_dec: text_lengths required");
  if (p->num_timesteps <= 0 || p->start_index < 0 || p->start_index >= p->num_timesteps)
    return fail(MDM_EINVAL, "mdm_sample_loop_dec: bad start_index / num_timesteps");
  if (!p->a_x0 || !p->a_xt || !p->sigma || !p->timestep_map) return fail(MDM_EINVAL, "mdm_sample_loop_dec: null schedule table");
  if ((p->inpaint_mask_dev == nullptr) != (p->inpaint_motion_dev == nullptr))
    return fail(MDM_EINVAL, "mdm_sample_loop_dec: inpainting needs mask and motion");
  const bool cfg = p->scale_dev != nullptr;
  const bool uncond_only = !cfg && (p->force_uncond || p->text_embed_dev == nullptr);
  if (cfg && p->text_embed_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop_dec: CFG needs the text tokens");
  if (p->num_dump > 0 && (p->dump_steps == nullptr || p->dump_dev == nullptr)) return fail(MDM_EINVAL, "mdm_sample_loop_dec: dump buffers missing");
  for (int i = 0; i <= p->start_index; ++i)
    if (p->timestep_map[i] < 0 || p->timestep_map[i] >= m->cfg.max_len) return fail(MDM_EINVAL, "mdm_sample_loop_dec: timestep outside the positional table");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int branches = cfg ? MDM_BRANCH_BOTH : (uncond_only ? MDM_BRANCH_UNCOND : MDM_BRANCH_COND);
  const int nbranch = cfg ? 2 : 1, nseq = nbranch * B, D = m->cfg.latent_dim, L = m->cfg.num_layers;
  const int nsteps = p->start_index + 1, Mm = nseq * ntok;
  DecWorkspace ws = carve_dec(m, nseq, m->cfg.context_len + P, ntok, B, ws_dev, nsteps, P);
  if (ws_bytes < ws.bytes) return fail(MDM_ENOSPC, "mdm_sample_loop_dec: workspace too small");
  const size_t per_sample = (size_t)m->jf * P;
  const bool x3 = m->precision == MDM_PREC_F16X3;
  Profiler* pf = &m->prof;

  // ---- once per window: what the steps share.  memory = embed_text(tokens) (cond) | bias (uncond)  +  time_emb[t]
  // (mdm.py:217-219, :262); its key | value projection of layer l is linear in the two parts:
  //   Wkv_l . memory + b = [Wkv_l . text part]  +  [Wkv_l . time_emb[t] + b]      (per token)   (per step)
  if (branches != MDM_BRANCH_UNCOND)
    if (int rc = launch_linear(nullptr, p->text_embed_dev, m->cfg.clip_dim, m->W("embed_text.weight"), m->W("embed_text.bias"),
                               nullptr, ws.proj, ntok * B, D, m->cfg.clip_dim, ACT_NONE, 0, 1.f, s)) return rc;
  {
    ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
    MDM_LAUNCH(text_memory_kernel, dim3(Mm), dim3(128), 0, s, ws.mem, (const float*)ws.proj, m->W("embed_text.bias"),
               (const float*)m->time_table, (const long long*)nullptr, B, ntok, D,
               (branches == MDM_BRANCH_UNCOND) ? 0 : 1, (int)m->cfg.max_len, m->time_add);
    if (int rc = rt_launch_status()) return rc;
  }
  for (int k0 = 0; k0 < nsteps; k0 += 64) {     // the steps' time-embedding rows: one gather launch per 64 steps
    RowGather g{};
    const int n = std::min(64, nsteps - k0);
    for (int k = 0; k < n; ++k) g.idx[k] = p->timestep_map[p->start_index - (k0 + k)];
    ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
    MDM_LAUNCH(gather_rows_kernel, dim3(n), dim3(128), 0, s, ws.time_rows + (size_t)k0 * D, (const float*)m->time_table, g, D);
    if (int rc = rt_launch_status()) return rc;
  }
  // ALL layers in one launch each (round 6; 2 L launches of 15 us before): rows of kv_text / kv_time are [L * 2D] wide, layer l at + l * 2D
  if (int rc = launch_linear(pf, ws.mem, D, m->wkv_all, nullptr, nullptr, ws.kv_text, Mm, L * 2 * D, D, ACT_NONE, 0, 1.f, s, x3)) return rc;
  if (int rc = launch_linear(pf, ws.time_rows, D, m->wkv_all, m->bkv_all, nullptr, ws.kv_time, nsteps, L * 2 * D, D, ACT_NONE, 0, 1.f, s, x3)) return rc;
  // sequence-tile route under guidance (csrc/decoder.h): the unconditional half's cross-attention block is one row constant per
  // sequence, layer and step -- Wo . (v_text + v_time) + bo -- linear in its two parts: both made here, once per window
  const bool o_hoist = cfg && x3 && dec_sequence_tiles(m, nseq, m->cfg.context_len + P);
  if (o_hoist)
    for (int l = 0; l < L; ++l) {
      const float* wo = m->L(l, "multihead_attn.out_proj.weight");
      const float* vt = ws.kv_text + (size_t)B * ntok * L * 2 * D + (size_t)l * 2 * D + D;     // first memory token of unconditional sequence 0
      if (int rc = launch_linear(pf, vt, ntok * L * 2 * D, wo, m->L(l, "multihead_attn.out_proj.bias"), nullptr,
                                 ws.o_text + (size_t)l * B * D, B, D, D, ACT_NONE, 0, 1.f, s)) return rc;
      if (int rc = launch_linear(pf, ws.kv_time + (size_t)l * 2 * D + D, L * 2 * D, wo, nullptr, nullptr,
                                 ws.o_time + (size_t)l * nsteps * D, nsteps, D, D, ACT_NONE, 0, 1.f, s)) return rc;
    }

  // ---- the steps.  The loop is written over G sample groups (each owns the rows [g * Mg, (g + 1) * Mg) of the activation
  // buffers and reads the hoisted text K / V of the whole batch through the attention kernel's (branch, sample) remap);
  // production runs ONE group on the caller's stream.
  // PROBE BUILD ONLY (MDM_DIP_GROUPS=G): the groups' loops run CONCURRENTLY on side streams (forked behind the hoisted
  // projections, joined before returning).  Samples are independent chains and a launch at these sizes is mostly fixed cost,
  // so one group's dispatch floor / cold loads / tail hide behind another's matrix work: +3 % (two groups) on the bench.
  // It is NOT in the product because in the f16x3 mode two or four concurrent chains intermittently (a few % of the window
  // loops at four groups, more with a split-precision attention kernel) return one sequence off by 1e-4 .. 1e-1 -- never in the
  // f32 mode, never with one chain, never with the groups serialised on one stream; not root-caused (profiles/r02e_dip.md,
  // reproducer tools/repro_dip_groups.py).
  int G = 1;
#ifdef MDM_PROBES
  {
    const char* e = getenv("MDM_DIP_GROUPS");
    const int want = e != nullptr ? atoi(e) : 1;
    for (int g = std::min(std::max(want, 1), AuxStreams::kMax + 1); g >= 1; --g)
      if (B % g == 0) { G = g; break; }
  }
#endif
  hipStream_t gs[AuxStreams::kMax + 1] = {s, s, s, s};
#if !defined(MDM_EMU) && defined(MDM_PROBES)
  if (G > 1) {
    if (int rc = m->aux.ensure(G - 1)) return rc;
    if (hipEventRecord(m->aux.fork, s) != hipSuccess) return fail(MDM_EHIP, "mdm_sample_loop_dec: hipEventRecord failed");
    for (int g = 1; g < G; ++g) {
      gs[g] = m->aux.s[g - 1];
      if (hipStreamWaitEvent(gs[g], m->aux.fork, 0) != hipSuccess) return fail(MDM_EHIP, "mdm_sample_loop_dec: hipStreamWaitEvent failed");
    }
  }
#endif
  const int Bg = B / G, nseq_g = nbranch * Bg, S = m->cfg.context_len + P;
  const size_t Mg = (size_t)nseq_g * S, FFs = m->cfg.ff_size;
  int dump_i = 0, k = 0, rc_loop = MDM_OK;
  for (int i = p->start_index; i >= 0 && rc_loop == MDM_OK; --i, ++k) {
    const bool dump = dump_i < p->num_dump && p->dump_steps[dump_i] == k;
    for (int g = 0; g < G && rc_loop == MDM_OK; ++g) {
      const int b0 = g * Bg;
      const size_t xo = (size_t)b0 * per_sample;
      DecWorkspace wg = ws;
      wg.tok += g * Mg * D; wg.qkv += g * Mg * 3 * D; wg.att += g * Mg * D; wg.ffn += g * Mg * FFs;
      wg.stat[0] += g * Mg * (D / LN_PART_COLS) * 2; wg.stat[1] += g * Mg * (D / LN_PART_COLS) * 2;
      wg.out += (size_t)g * nseq_g * per_sample;
      DecHoist hz;
      hz.step = k; hz.nsteps = nsteps; hz.kv_text = ws.kv_text; hz.kv_time = ws.kv_time; hz.kv_B = B; hz.kv_b0 = b0;
      hz.t_model = p->timestep_map[i];
      if (o_hoist) { hz.o_text = ws.o_text; hz.o_time = ws.o_time; }
      const float* prefix_g = pd->prefix_dev != nullptr ? pd->prefix_dev + (size_t)b0 * m->jf * m->cfg.context_len : nullptr;
      // CFG combine + posterior / DDIM update, in place on x (each element is read, then written, by the same lane): inside the
      // plane route's tail kernel (DecTail), else as a kernel of its own behind the denoiser
      StepCoefs co{p->a_x0[i], p->a_xt[i], p->sigma[i], p->clip_denoised};
      const float* step_noise = (p->noise_dev != nullptr && p->sigma[i] != 0.f) ? p->noise_dev + (size_t)k * B * per_sample + xo : nullptr;
      NoiseSource ns{step_noise, p->seed, p->sample_base + (uint32_t)b0, (uint32_t)(1 + k), (uint32_t)(p->const_noise != 0)};
      DecTail tail;
      tail.scale = cfg ? p->scale_dev + b0 : nullptr;
      tail.x = x + xo;
      tail.x0_out = (i == 0 && p->x0_dev != nullptr) ? p->x0_dev + xo : nullptr;
      tail.ns = ns;
      tail.inpaint_mask = p->inpaint_mask_dev != nullptr ? p->inpaint_mask_dev + xo : nullptr;
      tail.inpaint_motion = p->inpaint_motion_dev != nullptr ? p->inpaint_motion_dev + xo : nullptr;
      tail.co = co;
      rc_loop = decoder_pass(m, wg, x + xo, prefix_g, nullptr, p->text_embed_dev, pd->text_lengths_dev + b0,
                             p->lengths_dev, Bg, P, ntok, branches, wg.out, gs[g], hz, G == 1 ? &tail : nullptr);
      if (rc_loop != MDM_OK) break;
      const size_t total = (size_t)Bg * per_sample;
      const int grid = (int)std::min<size_t>((total + 255) / 256, 2048);
      if (!tail.done) {
        ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, gs[g]);
        MDM_LAUNCH(sampler_step_kernel, dim3(grid), dim3(256), 0, gs[g], (const float*)(x + xo), (const float*)wg.out,
                   cfg ? (const float*)(wg.out + (size_t)Bg * per_sample) : (const float*)nullptr,
                   cfg ? p->scale_dev + b0 : (const float*)nullptr,
                   p->inpaint_mask_dev != nullptr ? p->inpaint_mask_dev + xo : (const uint8_t*)nullptr,
                   p->inpaint_motion_dev != nullptr ? p->inpaint_motion_dev + xo : (const float*)nullptr, x + xo,
                   (i == 0 && p->x0_dev != nullptr) ? p->x0_dev + xo : (float*)nullptr, (int)per_sample, Bg, co, ns);
        rc_loop = rt_launch_status();
      }
      if (rc_loop == MDM_OK && dump)
        rc_loop = rt_copy(p->dump_dev + (size_t)dump_i * B * per_sample + xo, x + xo, (size_t)Bg * per_sample * sizeof(float), gs[g]);
    }
    if (dump) ++dump_i;
  }
#if !defined(MDM_EMU) && defined(MDM_PROBES)
  for (int g = 1; g < G; ++g)   // join, also on the error path: the caller's stream must not run ahead of the side streams
    if (hipEventRecord(m->aux.join[g - 1], gs[g]) != hipSuccess || hipStreamWaitEvent(s, m->aux.join[g - 1], 0) != hipSuccess)
      return fail(MDM_EHIP, "mdm_sample_loop_dec: joining the side streams failed");
#endif
  return rc_loop;
}
