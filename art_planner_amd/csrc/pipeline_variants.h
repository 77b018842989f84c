// pipeline_variants.h -- a form of the validity pipeline's foot stage that was built, pinned for parity and measured, and did
// NOT become the default (profiles/r05_lane_util.txt: VALU lane utilisation 0.50 -> 0.69, no time gained).  Only in
// libartp_variants.so (-DARTP_VARIANTS, selected there with $ARTP_FEET_DENSE=1).  Included from pipeline.h INSIDE namespace artp.
#pragma once

// ---- stage 1c, round 5: the same foot queue with the corner stage's expensive part on DENSE lanes ------------------
// feet_stream_kernel runs a box's corner stage on its 16 lanes: 8 corners x 2 triangle kinds nominate the cells under
// them, and the lanes whose triangle is KEPT form its plane, the box's contacts with it and the in-triangle tests
// (triangle_plane + box_plane_contacts + is_on_heightfield2, ~3/4 of the stage's instructions) -- typically 2-5 of the
// wavefront's 64 lanes, the others idle through it (SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU) = 0.50 for the kernel, the
// lowest of the pipeline; profiles/r05_lane_util.txt).  Here the nomination (gathers and compares: cheap, all lanes)
// pushes the kept candidates of ALL FOUR boxes of a round -- and of the following rounds -- into a per-wavefront LDS queue
// {queue item, cell, triangle kind, the cell's four samples}; whenever 64 are waiting, 64 lanes evaluate one each (the box
// pose comes back from the queue record: 48 bytes from the L2).  Verdicts are per box in LDS flags and leave at the end of
// the wavefront's chunk.  Same candidates, same arithmetic on the same operands as grp_corner_stage_direct: the same
// labels; only which lane computes what changes.
// MEASURED (profiles/r05_lane_util.txt): lane utilisation 0.50 -> 0.69, VALU instructions 2.20 -> 1.75 10^8 per batch,
// and 320.9 us against feet_stream_kernel's 308.3: no faster.  The SIMD executes a wave64 VALU instruction as four
// 16-lane passes and SKIPS a pass whose EXEC quarter is empty, so the 16-lane-group kernel's instructions cost ~3 cycles
// on average where these cost 4 (time x SIMDs / instructions) -- "idle lanes" in whole quarters are nearly free, and
// the per-instruction lane count overstates what re-packing can win.  Kept behind $ARTP_FEET_DENSE=1 with its parity test.
#define ARTP_FS2_QCAP 128
struct Fs2Cand {  // 24 bytes
  unsigned item;       // index into q.q1 (kept for diagnostics)
  unsigned cell;       // gx | gz << 16 (absolute cell of the triangle's A vertex)
  unsigned slot_kind;  // chunk slot of the box | c_up << 8
  float hA, hB, hC;
};
#ifndef ARTP_FEET2_WAVES_PER_SIMD
#define ARTP_FEET2_WAVES_PER_SIMD 6
#endif
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(ARTP_FEET2_WAVES_PER_SIMD, ARTP_FEET2_WAVES_PER_SIMD)))
feet_stream2_kernel(FieldDev ff, RobotDev rb, PipelineQueues q, uint8_t* __restrict__ valid) {
  constexpr int G = 16, GPW = 4, NBOX = ARTP_FEET_CHUNK;
  static_assert(NBOX <= 32, "one flag word per box, finalised by the first 32 lanes");
  __shared__ Fs2Cand cq_all[WAVES][ARTP_FS2_QCAP];
  __shared__ float cq_hd_all[WAVES][ARTP_FS2_QCAP];
  __shared__ unsigned box_item_all[WAVES][NBOX], box_state_all[WAVES][NBOX], box_flag_all[WAVES][NBOX];
  __shared__ float box_pose_all[WAVES][NBOX][12];   // pos[3], R[9] of the boxes that reached the corner stage
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Fs2Cand* cq = cq_all[wave];
  float* cq_hd = cq_hd_all[wave];
  unsigned* box_item = box_item_all[wave];
  unsigned* box_state = box_state_all[wave];
  float (*box_pose)[12] = box_pose_all[wave];
  unsigned* box_flag = box_flag_all[wave];   // bit 0 pending (reached the corner stage), bit 1 hit, bit 2 maybe-partner
  const int gl = lane & (G - 1);
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int sq = blockIdx.x % ARTP_NSUB;
  const unsigned long long count = *sub_counter(q, 1, sq);
  const unsigned long long first = sub_base(q, 1, sq);
  unsigned long long* cursor = sub_cursor(q, 1, sq);
  const float margin = 1.0e-4f;
  const float reach = 2.0f * margin * fmaxf(ff.inv_w, ff.inv_d);

  // n (<= 64) queued candidates from the head of the ring, one per lane
  int q_head = 0, n_q = 0;   // wave-uniform
  auto evaluate = [&](int n) {
    if (lane < n) {
      const int at = (q_head + lane) & (ARTP_FS2_QCAP - 1);
      const Fs2Cand c = cq[at];
      const float hD = cq_hd[at];
      const float* bp = box_pose[c.slot_kind & 0xffu];
      BoxHF b;
#pragma unroll
      for (int i = 0; i < 3; ++i) b.pos[i] = bp[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) b.R[i] = bp[3 + i];
      b.side[0] = rb.foot[0]; b.side[1] = rb.foot[1]; b.side[2] = rb.foot[2];
      const bool c_up = ((c.slot_kind >> 8) & 1u) != 0;
      const int gx0 = (int)(c.cell & 0xffffu), gz0 = (int)(c.cell >> 16);
      const float xA = (float)gx0 * ff.sample_w, xB = (float)(gx0 + 1) * ff.sample_w;
      const float zA = (float)gz0 * ff.sample_d, zC = (float)(gz0 + 1) * ff.sample_d;
      float cpl[4];
      if (c_up)
        triangle_plane(xA, c.hA, zA, xB, c.hB, zA, xA, c.hC, zC, true, cpl);
      else
        triangle_plane(xB, hD, zC, xB, c.hB, zA, xA, c.hC, zC, false, cpl);
      const int gx = gx0 + (c_up ? 0 : 1), gz = gz0 + (c_up ? 0 : 1);
      float cpos[4][3];
      const int nc = box_plane_contacts(b, cpl[0], cpl[1], cpl[2], cpl[3], 10, cpos);
      bool hit = false;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nc) hit = hit || is_on_heightfield2(ff, gx, gz, cpos[i][0], cpos[i][2], c_up);
      if (hit) atomicOr(&box_flag[c.slot_kind & 0xffu], 2u);
    }
    q_head = (q_head + n) & (ARTP_FS2_QCAP - 1);
    n_q -= n;
  };

  for (;;) {
    unsigned long long chunk = 0;
    if (lane == 0) chunk = atomicAdd(cursor, (unsigned long long)NBOX);
    chunk = __shfl(chunk, 0);
    if (chunk >= count) break;
    if (lane < NBOX) box_flag[lane] = 0u;
    wave_lds_sync();
    for (int r = 0; r < NBOX / GPW; ++r) {
      const int slot = r * GPW + lane / G;
      const unsigned long long it = chunk + (unsigned long long)slot;
      bool need = false;
      BoxHF b;
      unsigned item = 0;
      if (it < count) {
        item = (unsigned)(first + it);
        const PendingBox rec = q.q1[item];
        if (valid[rec.state] != 0 && (rec.kind & ARTP_REC_EXITS_NEGATIVE)) {
          box_from_record(rec, rb, b);
          const bool touches = grp_vertex_stream<G, ARTP_FEET_U>(ff, b, lane, (rec.kind & ARTP_REC_ALL_FINITE) != 0);
          need = !touches;
          if (need) {
#pragma unroll
            for (int i = 0; i < 12; ++i)   // static register indices (a dynamic one would put the box in scratch)
              if (gl == i) box_pose[slot][i] = i < 3 ? b.pos[i] : b.R[i - 3];
            if (gl == 0) {
              box_item[slot] = item;
              box_state[slot] = rec.state;
              box_flag[slot] = 1u;
            }
          }
        }
      }
      if (!__any(need)) continue;   // wave-uniform
      // nomination: grp_corner_stage_direct up to its `kept`, every lane of the wavefront in step
      int cellsX = 0, cellsZ = 0, cxa = 0, cxb = -1, cza = 0, czb = -1;
      float minO2 = 0.f, py = 0.f;
      bool window_covered = false, c_up = false;
      if (need) {
        cellsX = b.maxX - b.minX;
        cellsZ = b.maxZ - b.minZ;
        minO2 = b.aabb[2];
        window_covered = ff.partner_flags != nullptr && cellsX <= ff.partner_R && cellsZ <= ff.partner_R;
        const int base_slot = gl & 15;
        const int corner = base_slot >> 1;
        c_up = !(base_slot & 1);
        const float s0 = (corner & 1) ? 0.5f : -0.5f, s1 = (corner & 2) ? 0.5f : -0.5f, s2 = (corner & 4) ? 0.5f : -0.5f;
        const float px = b.pos[0] + s0 * b.side[0] * b.R[0] + s1 * b.side[1] * b.R[1] + s2 * b.side[2] * b.R[2];
        py = b.pos[1] + s0 * b.side[0] * b.R[3] + s1 * b.side[1] * b.R[4] + s2 * b.side[2] * b.R[5];
        const float pz = b.pos[2] + s0 * b.side[0] * b.R[6] + s1 * b.side[1] * b.R[7] + s2 * b.side[2] * b.R[8];
        cxa = (int)floorf((px - margin) * ff.inv_w);
        cxb = (int)floorf((px + margin) * ff.inv_w);
        cza = (int)floorf((pz - margin) * ff.inv_d);
        czb = (int)floorf((pz + margin) * ff.inv_d);
      }
      bool maybe_partner = false;
      for (int o = 0; o < 4; ++o) {
        const int dx = o & 1, dz = o >> 1;
        const bool wanted = need && (cxa + dx <= cxb) && (cza + dz <= czb);
        if (!__any(wanted)) continue;   // wave-uniform
        bool kept = false;
        float hA = 0.f, hB = 0.f, hC = 0.f, hD = 0.f;
        int cx = 0, cz = 0;
        if (wanted) {
          cx = cxa + dx - b.minX;
          cz = cza + dz - b.minZ;  // window-local cell
          if (cx >= 0 && cz >= 0 && cx < cellsX && cz < cellsZ) {
            const unsigned at = (unsigned)((b.minX + cx) + (b.minZ + cz) * ff.nW);
            hA = gather32(ff.data, at);
            hB = gather32(ff.data, at + 1u);
            hC = gather32(ff.data, at + (unsigned)ff.nW);
            hD = gather32(ff.data, at + (unsigned)ff.nW + 1u);
            const bool fA = is_finite(hA), fB = is_finite(hB), fC = is_finite(hC), fD = is_finite(hD);
            const bool kA = fA && hA > minO2, kB = fB && hB > minO2, kC = fC && hC > minO2, kD = fD && hD > minO2;
            kept = c_up ? ((kA || kB || kC) && (fA && fB && fC)) : ((kB || kC || kD) && (fB && fC && fD));
            if (kept) {
              const float h0 = c_up ? hA : hD;
              const float h4 = (hB + hC) - h0;
              const float top = fmaxf(fmaxf(h0, h4), fmaxf(hB, hC));
              const float spread = fabsf(hB - h0) + fabsf(hC - h0);
              kept = !(py > top + spread * reach + 1.0e-3f);
            }
            if (kept)
              maybe_partner = maybe_partner || !window_covered || ((gather32(ff.partner_flags, at) >> (c_up ? 0 : 1)) & 1);
          }
        }
        const unsigned long long kb = __ballot(kept);
        if (kb) {   // wave-uniform
          const int nk = __popcll(kb);
          if (n_q + nk > ARTP_FS2_QCAP) {
            // more kept candidates in one round than the queue holds (it starts a round below 64 and a round of four
            // boxes typically adds ~20): their boxes go to the list pass (queue 5) like boxes with a possible partner --
            // it evaluates the whole corner stage itself.  No evaluation in here: nothing of the nomination is live
            // across the matrix of registers box_plane_contacts wants.
            if (kept) atomicOr(&box_flag[slot], 4u);
          } else {
            if (kept) {
              const int at_q = (q_head + n_q + __popcll(kb & lt_mask)) & (ARTP_FS2_QCAP - 1);
              Fs2Cand c;
              c.item = item;
              c.cell = (unsigned)(b.minX + cx) | ((unsigned)(b.minZ + cz) << 16);
              c.slot_kind = (unsigned)slot | (c_up ? 256u : 0u);
              c.hA = hA;
              c.hB = hB;
              c.hC = hC;
              cq[at_q] = c;
              cq_hd[at_q] = hD;
            }
            n_q += nk;
          }
        }
      }
      if (grp_any<G>(maybe_partner, lane) && need && gl == 0) atomicOr(&box_flag[slot], 4u);
      // keep the queue short: whenever a full wavefront of candidates waits, it goes through
      while (n_q >= 64) {
        wave_lds_sync();
        evaluate(64);
      }
    }
    wave_lds_sync();
    if (n_q > 0) evaluate(n_q);
    wave_lds_sync();
    if (lane < NBOX) {
      const unsigned fl = box_flag[lane];
      if (fl & 1u) {
        if (fl & 4u)
          q.q5[atomicAdd(&q.counters[6], 1ull)] = box_item[lane];
        else if (!(fl & 2u))
          valid[box_state[lane]] = 0;  // a foot that touches nothing fails the state
      }
    }
    wave_lds_sync();
  }
}
