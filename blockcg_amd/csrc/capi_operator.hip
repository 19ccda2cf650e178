// include/blockcg_hip.h, part 2 of 3: the operator -- halo exchange, the stencil launches (kernels_stencil.hip), dirac_op::op
// (inc/dirac_op.hpp:36-43) with a whole `tmp`, as capacity mode's ring sweep and on half-volume fields, and the gauge API.
#include "capi_internal.hpp"

namespace bcg_impl {

// ---- halo exchange -----------------------------------------------------------------------------
// rank = lexicographic index of grid coordinates, direction 0 fastest
int rank_of_grid(const int* grid, const int* xyz) {
  int r = 0, st = 1;
  for (int mu = 0; mu < 4; ++mu) {
    r += xyz[mu] * st;
    st *= grid[mu];
  }
  return r;
}

// The message plan of one halo exchange (pure host arithmetic, shared by bcg_halo_plan and the
// context).  Ghost/send buffers hold, per split direction in ascending mu, [minus face][plus face]
// (send buffer: [low face x_mu = 0][high face x_mu = L-1]); a face has V_local / L_mu sites.
//   message 2k  : low face  -> minus neighbour (becomes its plus ghost); my plus ghost  <- plus neighbour
//   message 2k+1: high face -> plus neighbour  (becomes its minus ghost); my minus ghost <- minus neighbour
int halo_plan(int ndim, const int* gdims, const int* grid, const int* coords, size_t site_bytes, int* peer_s, int* peer_r,
              size_t* off_s, size_t* off_r, size_t* nb, int64_t* ghost_sites) {
  int g4[4] = {1, 1, 1, 1}, c4[4] = {0, 0, 0, 0}, L[4] = {1, 1, 1, 1};
  int64_t V = 1;
  for (int mu = 0; mu < ndim; ++mu) {
    g4[mu] = grid ? grid[mu] : 1;
    c4[mu] = coords ? coords[mu] : 0;
    if (g4[mu] < 1 || gdims[mu] < 1 || gdims[mu] % g4[mu] != 0 || c4[mu] < 0 || c4[mu] >= g4[mu]) return -1;
    L[mu] = gdims[mu] / g4[mu];
    V *= L[mu];
  }
  int n = 0;
  int64_t ghost = 0;
  for (int mu = 0; mu < ndim; ++mu) {
    if (g4[mu] == 1) continue;
    int xm[4], xp[4];
    for (int nu = 0; nu < 4; ++nu) xm[nu] = xp[nu] = c4[nu];
    xm[mu] = (c4[mu] - 1 + g4[mu]) % g4[mu];
    xp[mu] = (c4[mu] + 1) % g4[mu];
    const int rm = rank_of_grid(g4, xm), rp = rank_of_grid(g4, xp);
    const int64_t face_sites = V / L[mu];
    const size_t face = static_cast<size_t>(face_sites) * site_bytes;
    const size_t base = static_cast<size_t>(ghost) * site_bytes;
    peer_s[n] = rm; peer_r[n] = rp; off_s[n] = base; off_r[n] = base + face; nb[n] = face; ++n;
    peer_s[n] = rp; peer_r[n] = rm; off_s[n] = base + face; off_r[n] = base; nb[n] = face; ++n;
    ghost += 2 * face_sites;
  }
  if (ghost_sites) *ghost_sites = ghost;
  return n;
}

// Post the face messages for `site_bytes` bytes per site (fields: 3*m*16; gauge: 9*16).  split = true uses the
// begin half of the optional split form (the caller then issues exchange_end after the interior tiles).
// x3_n > 0 (direction 3 undivided): only the slices [x3_lo, x3_lo + x3_n), a contiguous sub-range of every face.
// x3b_n > 0: a second range of slices in the same exchange (the messages of the first range, then those of the second)
int exchange_faces(bcg_context* c, size_t site_bytes, bool split = false, int x3_lo = 0, int x3_n = 0, int x3b_lo = 0, int x3b_n = 0) {
  if (!c->have_comm || !c->comm.halo_exchange) BCG_FAIL(c, BCG_ERR_COMM, "lattice is split over ranks but no bcg_comm was set");
  int peer_s[16], peer_r[16];
  size_t off_s[16], off_r[16], nb[16];
  int n = halo_plan(c->ndim, c->gdims, c->grid, c->coords, site_bytes, peer_s, peer_r, off_s, off_r, nb, nullptr);
  if (n < 0) BCG_FAIL(c, BCG_ERR_INVALID, "halo plan");
  if (x3_n > 0) {
    for (int k = 0; k < n; ++k) {
      const size_t slice = nb[k] / c->lat.L[3];
      if (x3b_n > 0) {
        peer_s[n + k] = peer_s[k];
        peer_r[n + k] = peer_r[k];
        off_s[n + k] = off_s[k] + slice * x3b_lo;
        off_r[n + k] = off_r[k] + slice * x3b_lo;
        nb[n + k] = slice * x3b_n;
      }
      off_s[k] += slice * x3_lo;
      off_r[k] += slice * x3_lo;
      nb[k] = slice * x3_n;
    }
    if (x3b_n > 0) n *= 2;
  }
  auto fn = split ? c->comm.halo_exchange_begin : c->comm.halo_exchange;
  if (fn(c->comm.user, n, peer_s, peer_r, off_s, off_r, nb) != 0) BCG_FAIL(c, BCG_ERR_COMM, "halo_exchange callback failed");
  return BCG_OK;
}
int exchange_end(bcg_context* c) {
  if (c->comm.halo_exchange_end(c->comm.user) != 0) BCG_FAIL(c, BCG_ERR_COMM, "halo_exchange_end callback failed");
  return BCG_OK;
}

int halo_field(bcg_context* c, const bcg_field* f, bool split) {
  if (!c->distributed) return BCG_OK;
  const size_t site_bytes = static_cast<size_t>(3) * f->m * sizeof(double2);
  BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
  if (f->parity >= 0) {
    // a half-volume field: every face holds half its sites (kernels_generic.hip: k_pack_faces_half), at half the offsets
    // of the full plan -- the same messages with half the bytes per site (face sizes are even: every extent is)
    {
      ProfScope ps(c, "pack_faces");
      bcg::launch_pack_faces_half(c->stream, f->m, c->lat, f->parity, f->d, c->halo_send);
    }
    BCG_TRY(check_launch(c, "pack_faces"));
    ProfScope ps(c, split ? "halo_exchange_begin" : "halo_exchange");
    return exchange_faces(c, site_bytes / 2, split);
  }
  {
    ProfScope ps(c, "pack_faces");
    bcg::launch_pack_faces(c->stream, f->m, c->lat, f->d, c->halo_send);
  }
  BCG_TRY(check_launch(c, "pack_faces"));
  ProfScope ps(c, split ? "halo_exchange_begin" : "halo_exchange");
  return exchange_faces(c, site_bytes, split);
}

// Faces of the x3 slices [x3_lo, x3_lo + x3_n) only; `d` is a whole field (ring = 0) or a ring of slices (capacity mode).
// The other slices' ranges of the ghost buffer keep what they held.
// x3b_n > 0: and those of a second range of slices, in the same exchange
// parity >= 0: `d` is a half-volume field of that parity (whole, ring = 0): half faces, half the bytes per site (halo_field)
int halo_window(bcg_context* c, int m, const double2* d, int x3_lo, int x3_n, int ring, bool split = false, int x3b_lo = 0,
                int x3b_n = 0, int parity = -1) {
  if (!c->distributed) return BCG_OK;
  const size_t site_bytes = static_cast<size_t>(3) * m * sizeof(double2);
  BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
  {
    ProfScope ps(c, "pack_faces");
    if (parity >= 0) {
      bcg::launch_pack_faces_half(c->stream, m, c->lat, parity, d, c->halo_send, x3_lo, x3_n);
      if (x3b_n > 0) bcg::launch_pack_faces_half(c->stream, m, c->lat, parity, d, c->halo_send, x3b_lo, x3b_n);
    } else {
      bcg::launch_pack_faces(c->stream, m, c->lat, d, c->halo_send, x3_lo, x3_n, ring);
      if (x3b_n > 0) bcg::launch_pack_faces(c->stream, m, c->lat, d, c->halo_send, x3b_lo, x3b_n, ring);
    }
  }
  BCG_TRY(check_launch(c, "pack_faces"));
  ProfScope ps(c, split ? "halo_exchange_begin" : "halo_exchange");
  return exchange_faces(c, parity >= 0 ? site_bytes / 2 : site_bytes, split, x3_lo, x3_n, x3b_lo, x3b_n);
}

// Capacity mode with overlapped exchanges: the received faces of slice x3 = 0 of every split direction, saved aside
// (save) or put back (!save).  The ghost ranges of slice 0 are re-used for the faces of `tmp` while the source's faces of
// that slice are needed once more at the end of the sweep (apply_shifted_ring).
int ensure_halo_save(bcg_context* c, size_t total) {
  if (total <= c->halo_save_bytes) return BCG_OK;
  BCG_TRY(stream_sync(c));
  if (c->halo_save) (void)hipFree(c->halo_save);
  c->halo_save = nullptr;
  c->halo_save_bytes = 0;
  HIP_TRY(c, hipMalloc(&c->halo_save, total));
  c->halo_save_bytes = total;
  return BCG_OK;
}
int slice0_faces(bcg_context* c, size_t site_bytes, bool save) {
  int peer_s[8], peer_r[8];
  size_t off_s[8], off_r[8], nb[8];
  const int n = halo_plan(c->ndim, c->gdims, c->grid, c->coords, site_bytes, peer_s, peer_r, off_s, off_r, nb, nullptr);
  if (n < 0) BCG_FAIL(c, BCG_ERR_INVALID, "halo plan");
  size_t total = 0;
  for (int k = 0; k < n; ++k) total += nb[k] / c->lat.L[3];
  BCG_TRY(ensure_halo_save(c, total));
  size_t at = 0;
  for (int k = 0; k < n; ++k) {
    const size_t each = nb[k] / c->lat.L[3];
    char* const ghost = reinterpret_cast<char*>(c->halo_recv) + off_r[k];
    char* const keep = reinterpret_cast<char*>(c->halo_save) + at;
    HIP_TRY(c, hipMemcpyAsync(save ? keep : ghost, save ? ghost : keep, each, hipMemcpyDeviceToDevice, c->stream));
    at += each;
  }
  return BCG_OK;
}

int halo_gauge(bcg_context* c, bcg_gauge* g) {
  if (!c->distributed || g->ghost_valid) return BCG_OK;
  const size_t site_bytes = 9 * sizeof(double2);
  BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
  bcg::launch_pack_gauge_faces(c->stream, c->lat, g->U, c->halo_send);
  BCG_TRY(check_launch(c, "pack_gauge_faces"));
  BCG_TRY(exchange_faces(c, site_bytes));
  HIP_TRY(c, hipMemcpyAsync(g->Ughost, c->halo_recv, static_cast<size_t>(c->ghost_sites) * site_bytes,
                            hipMemcpyDeviceToDevice, c->stream));
  BCG_TRY(stream_sync(c));
  g->ghost_valid = true;
  return BCG_OK;
}

// The boundary tiles (a site of the tile has a neighbour in a ghost face) of the tiling with `spb` sites per tile, in
// lexicographic order; built once per tile length.  The boundary launch deals them to its blocks round-robin.
int boundary_tile_list(bcg_context* c, int spb, const int** list, int* n) {
  auto it = c->boundary_tiles.find(spb);
  if (it == c->boundary_tiles.end()) {
    const bcg::LatticeDev& L = c->lat;
    std::vector<int> tiles;
    for (int x3 = 0; x3 < L.L[3]; ++x3)
      for (int x2 = 0; x2 < L.L[2]; ++x2)
        for (int x1 = 0; x1 < L.L[1]; ++x1) {
          const bool b123 = (L.split[1] && (x1 == 0 || x1 == L.L[1] - 1)) || (L.split[2] && (x2 == 0 || x2 == L.L[2] - 1)) ||
                            (L.split[3] && (x3 == 0 || x3 == L.L[3] - 1));
          for (int x0b = 0; x0b < L.L[0]; x0b += spb)
            if (b123 || (L.split[0] && (x0b == 0 || x0b + spb == L.L[0])))
              tiles.push_back(x0b + L.L[0] * (x1 + L.L[1] * (x2 + L.L[2] * x3)));
        }
    int* dev = nullptr;
    if (!tiles.empty()) {
      HIP_TRY(c, hipMalloc(&dev, tiles.size() * sizeof(int)));
      HIP_TRY(c, hipMemcpy(dev, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    it = c->boundary_tiles.emplace(spb, std::make_pair(dev, static_cast<int>(tiles.size()))).first;
  }
  *list = it->second.first;
  *n = it->second.second;
  return BCG_OK;
}

// Profiling only: count the launches of each form of the stencil kernel ("stencil_form_k_hop4c" ...), so that tests
// and tuning runs can tell which one a lattice shape gets.
void note_stencil_form(bcg_context* c, int m, int tile_class, const bcg::HopWindow& win, bool plain = false) {
  if (!c->profiling) return;
  static const char* names[] = {"stencil_form_general", "stencil_form_k_hop4", "stencil_form_k_hop4c", "stencil_form_k_hop4b"};
  int form = bcg::hop_kernel_form(m, c->lat, kFastBlocks, c->hop_tune, tile_class, win);
  if (form == 2 && bcg::hop_uses_bundle(m, c->lat, kFastBlocks, c->hop_tune, tile_class, win, plain)) form = 3;
  if (form >= 0 && form <= 3) c->prof[names[form]].count += 1;
}

// out = D in  (HOP_PLAIN)  or  out = c0*p - D in  (HOP_SHIFTED).  With gram_blocks != nullptr (m = 16 fast
// path, HOP_SHIFTED) the kernel also leaves block partials of p^dagger out in c->partials.
int hop(bcg_context* c, const bcg_gauge* g, bcg_field* out, const bcg_field* in, bcg::HopMode mode, const bcg_field* p,
        double c0, int* gram_blocks, bool* gram_folded) {
  if (in->parity >= 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "D alone maps a half-volume field to the other parity: use bcg_dirac_hop_half");
  BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
  const int m = in->m;
  if (gram_blocks) *gram_blocks = 0;
  if (gram_folded) *gram_folded = false;
  const bool fast = fast_hop(c, m);
  // fused Gram product: m = 16 in every form of the specialised stencil; m = 8 in the column-sweep kernel, one launch
  const bool split_path = fast && bcg::hop_can_split_tiles(m, c->lat) && can_overlap(c);
  const bool gram = fast && gram_blocks && mode == bcg::HOP_SHIFTED &&
                    (m == 16 || (m == 8 && !split_path &&
                                 bcg::hop_kernel_form(m, c->lat, kFastBlocks, c->hop_tune, 0, bcg::HopWindow()) == 2));
  const char* name = gram ? "hop_shifted_gram" : (mode == bcg::HOP_PLAIN ? "hop" : "hop_shifted");
  if (fast) BCG_TRY(ensure_scratch(c));
  // BCG_FORCE_TILE_CLASSES=1 (tuning aid): take the two-launch path on an undivided lattice too, where every tile is
  // an interior one, to time the interior-class kernel on one GPU
  const bool force_classes = c->force_tile_classes;
  if (fast && bcg::hop_can_split_tiles(m, c->lat) && (can_overlap(c) || (force_classes && !c->distributed))) {
    // pack -> post the exchange -> interior tiles (no ghost reads) -> wait for the exchange -> boundary tiles
    if (c->distributed) BCG_TRY(halo_field(c, in, /*split=*/true));
    bcg::HopTuning tune = c->hop_tune;
    tune.blocks = tune.blocks_overlap;  // leave some CUs to the transport's kernels while it runs
    int nb1, nb2;
    note_stencil_form(c, m, 1, bcg::HopWindow());
    {
      ProfScope ps(c, name, alg_bytes(c, m, mode == bcg::HOP_PLAIN ? 2 : 3, 1), hop_flops(c, m, gram));  // both tile classes: counted here
      nb1 = bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                                 p ? p->d : nullptr, c0, c->partials, gram, kFastBlocks, tune, /*interior*/ 1);
    }
    BCG_TRY(check_launch(c, name));
    if (c->distributed) {
      ProfScope ps(c, "halo_exchange_end");
      BCG_TRY(exchange_end(c));
    }
    {
      bcg::HopTuning tb = c->hop_tune;
      BCG_TRY(boundary_tile_list(c, 4 * (64 / m), &tb.boundary_list, &tb.boundary_n));
      if (tb.boundary_n == 0) tb.boundary_list = nullptr;  // nothing to do: fall through to an empty class launch
      ProfScope ps(c, "hop_boundary");
      nb2 = tb.boundary_n == 0 ? 0
                               : bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                                                      p ? p->d : nullptr, c0,
                                                      gram ? c->partials + static_cast<size_t>(nb1) * m * m : c->partials, gram,
                                                      kFastBlocks, tb, /*boundary*/ 2);
    }
    if (gram) *gram_blocks = nb1 + nb2;
    return check_launch(c, "hop_boundary");
  }
  BCG_TRY(halo_field(c, in));
  if (fast) {
    note_stencil_form(c, m, 0, bcg::HopWindow(), mode == bcg::HOP_PLAIN);
    bcg::HopTuning tune = c->hop_tune;
    // one whole launch of a column form: the kernel's last blocks sum the Gram partials themselves (no reduction launch)
    const bool fold = gram && gram_folded && bcg::hop_folds_gram(m, c->lat, kFastBlocks, tune, bcg::HopWindow());
    if (fold) tune.fold = bcg::GramFold{c->dev_gram, c->fold_tickets};
    ProfScope ps(c, name, alg_bytes(c, m, mode == bcg::HOP_PLAIN ? 2 : 3, 1), hop_flops(c, m, gram));
    const int nb = bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                                        p ? p->d : nullptr, c0, c->partials, gram, kFastBlocks, tune, 0);
    if (gram) *gram_blocks = nb;
    if (fold) *gram_folded = true;
  } else {
    ProfScope ps(c, name, alg_bytes(c, m, mode == bcg::HOP_PLAIN ? 2 : 3, 1), hop_flops(c, m, gram));
    bcg::launch_hop_generic(c->stream, m, c->lat, g->U, g->Ughost, in->d, c->halo_recv, out->d, mode,
                            p ? p->d : nullptr, c0);
  }
  return check_launch(c, "hop");
}

int get_tmp(bcg_context* c, int m, bcg_field** out) {
  auto it = c->tmp_field.find(m);
  if (it != c->tmp_field.end()) {
    *out = it->second;
    return BCG_OK;
  }
  bcg_field* f = nullptr;
  BCG_TRY(bcg_field_create(c, m, &f));
  c->tmp_field[m] = f;
  *out = f;
  return BCG_OK;
}
// the half-volume `tmp` of parity `parity` (= D applied to a field of the other parity)
int get_tmp_half(bcg_context* c, int m, int parity, bcg_field** out) {
  const int key = m + 1000 * (1 + parity);
  auto it = c->tmp_field.find(key);
  if (it != c->tmp_field.end()) {
    *out = it->second;
    return BCG_OK;
  }
  bcg_field* f = nullptr;
  BCG_TRY(bcg_field_create_half(c, m, parity, &f));
  c->tmp_field[key] = f;
  *out = f;
  return BCG_OK;
}

// Capacity mode (bcg_capacity_mode): the same T = (mass^2 + sigma0) P - D(D(P)), with tmp = D P held as a ring of R x3
// slices instead of a whole field.  Direction 3 is undivided, so a slice of T needs the slices x3-1, x3, x3+1 of tmp and
// nothing else of it: the first stencil runs C = R - 2 slices ahead of the second.
//   tmp[L3-1]; then per chunk [lo, hi) of C slices: tmp[.. hi] (slice L3 = slice 0 again), faces of tmp[lo, hi) to the
//   neighbours, T[lo, hi).  Writing slice s of tmp replaces slice s - R, which no later chunk reads.
// Slices L3-1 and 0 of tmp are computed twice (2/L3 more work in the first stencil).  The ghost buffer is shared: the faces of
// tmp[lo, hi) land on the range that held the faces of P[lo, hi), which the first stencil no longer reads -- except
// slice 0 at the very end, whose P faces are exchanged again (serial form) or restored from a copy (overlapped form, below).
bool capacity_path(const bcg_context* c, int m) {
  return c->tmp_ring > 0 && fast_hop(c, m) && bcg::hop_can_split_tiles(m, c->lat);
}
inline int ring_chunk(const bcg_context* c) {
  const int most = ring_overlapped(c) ? (c->tmp_ring - 2) / 2 : c->tmp_ring - 2;
  // BCG_RING_CHUNK (tests, tuning): shorter chunks than the ring allows -- e.g. the overlapped form's 15-slice windows of
  // ring 32 on a single rank, where the serial form would sweep 30 slices at a time
  return c->ring_chunk_override > 0 && c->ring_chunk_override < most ? c->ring_chunk_override : most;
}
// Everything capacity mode allocates for width m: the ring, the block partials of all chunks side by side (the stencil
// grid stays the tuned one: a smaller grid loses the x3 walk), the face buffers and the copy of the slice-0 faces.
int ensure_ring_scratch(bcg_context* c, int m) {
  const int R = c->tmp_ring, L3 = c->lat.L[3], C = ring_chunk(c);
  double2*& ring = c->tmp_ring_buf[m];
  if (!ring) HIP_TRY(c, hipMalloc(&ring, static_cast<size_t>(R) * c->lat.stride[3] * 3 * m * sizeof(double2)));
  BCG_TRY(ensure_scratch(c));
  const int chunks = (L3 + C - 1) / C;
  const size_t need = static_cast<size_t>(c->hop_tune.blocks > 0 ? c->hop_tune.blocks : kFastBlocks) * chunks * m * m * sizeof(double2);
  if (m == 16 && need > c->partials_bytes) {
    BCG_TRY(stream_sync(c));
    (void)hipFree(c->partials);
    c->partials = nullptr;
    c->partials_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->partials, need));
    c->partials_bytes = need;
  }
  if (c->distributed) {
    const size_t site_bytes = static_cast<size_t>(3) * m * sizeof(double2);
    BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * site_bytes));
    if (ring_overlapped(c)) BCG_TRY(ensure_halo_save(c, static_cast<size_t>(c->ghost_sites) / L3 * site_bytes));
  }
  return BCG_OK;
}
// Half-volume fields on a lattice divided over ranks, direction 3 undivided: the same sweep in chunks of x3 slices on a WHOLE
// tmp (half field; no ring), for the sake of its overlapped exchanges -- the faces of the source in two windows, those of tmp
// chunk by chunk, each travelling while the neighbouring chunks are computed (apply_shifted_ring with half_tmp set).
inline int half_chunk(const bcg_context* c) { return c->half_chunk_override > 0 ? c->half_chunk_override : 16; }
inline bool half_chunked_path(const bcg_context* c) {
  // (half_chunk_force: BCG_HALF_CHUNK_FORCE=1, a tuning aid -- the chunked sweep on one GPU, to time what the chunks cost)
  return ((c->distributed && can_overlap(c)) || c->half_chunk_force) && c->ndim == 4 && !c->lat.split[3] && !c->lat.split[0] &&
         c->lat.L[3] > half_chunk(c);
}
int ensure_half_chunk_scratch(bcg_context* c, int m) {
  BCG_TRY(ensure_scratch(c));
  const int C = half_chunk(c), chunks = (c->lat.L[3] + C - 1) / C;
  const size_t need = static_cast<size_t>(c->hop_tune.blocks > 0 ? c->hop_tune.blocks : kFastBlocks) * chunks * m * m * sizeof(double2);
  if (m == 16 && need > c->partials_bytes) {  // the block partials of all chunks side by side, as in capacity mode
    BCG_TRY(stream_sync(c));
    (void)hipFree(c->partials);
    c->partials = nullptr;
    c->partials_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->partials, need));
    c->partials_bytes = need;
  }
  return ensure_halo(c, static_cast<size_t>(c->ghost_sites) * 3 * m * sizeof(double2));
}
int apply_shifted_ring(bcg_context* c, const bcg_gauge* g, double mass, double sigma0, bcg_field* T, const bcg_field* P,
                       int* gram_blocks, bcg_field* half_tmp = nullptr) {
  const bool half = half_tmp != nullptr;  // P, T: half fields of one parity, half_tmp: the whole tmp of the other
  const int m = P->m, L3 = c->lat.L[3], R = half ? L3 : c->tmp_ring;
  // Overlapped form (ranks that exchange faces, split callbacks present, ring of at least 2 C + 2 slices): the exchange of
  // chunk k's tmp faces runs while the first stencil works on chunk k + 1 and the second one on chunk k - 1, so the ring
  // holds two chunks and the two boundary slices.  Otherwise C = R - 2 and every exchange is waited for where it is posted.
  const bool overlap = half ? can_overlap(c) : ring_overlapped(c);
  const int C = half ? half_chunk(c) : ring_chunk(c);
  BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
  if (half) BCG_TRY(ensure_half_chunk_scratch(c, m));
  else BCG_TRY(ensure_ring_scratch(c, m));
  double2* const ring = half ? half_tmp->d : c->tmp_ring_buf[m];
  if (gram_blocks) *gram_blocks = 0;
  const bool gram = gram_blocks && m == 16;
  const bcg::HopTuning& tune = c->hop_tune;
  const size_t site_bytes = static_cast<size_t>(3) * m * sizeof(double2) / (half ? 2 : 1);  // (of the face messages)
  // half fields: the compact lattice with the half ghost faces' offsets (apply_shifted), windows without ring addressing
  bcg::LatticeDev lat = c->lat;
  if (half) {
    lat.L[0] /= 2;
    lat.V /= 2;
    for (int mu = 1; mu < 4; ++mu) lat.stride[mu] /= 2;
    for (int mu = 0; mu < 4; ++mu) {
      lat.face_sites[mu] /= 2;
      lat.ghost_off[mu][0] /= 2;
      lat.ghost_off[mu][1] /= 2;
    }
  }
  const int par_p = half ? P->parity : -1, par_t = half ? half_tmp->parity : -1;
  const int vden = half ? 2 * L3 : L3;  // a window's share of the full local volume
  // Algorithmic link bytes of a launch on HALF fields: every output site needs its four forward links AND the four backward
  // links U_mu(x - mu), which live at sites of the other parity and are nobody's forward link in this launch -- a half-volume
  // stencil reads ALL the lattice's links, 2 x 576 B per output site, where the full-volume one reads each link once for two
  // uses.  (Rounds 3-4 priced one pass: the checkerboard form's "0.44 / 0.49 of the HBM peak" was 27 % / 20 % under-priced.)
  const double kLinkPasses = half ? 2.0 : 1.0;
  auto window = [&](int lo, int n, int parity_out) {
    bcg::HopWindow w;
    w.x3_lo = lo;
    w.x3_n = n;
    w.ring = half ? 0 : R;
    w.cb = half ? 1 : 0;
    w.cb_parity = half ? parity_out : 0;
    return w;
  };
  // The source's faces.  Serial form: one blocking exchange of the whole field.  Overlapped form: nothing blocks -- the
  // faces of the slices the first launches read (the wrap slice L3 - 1 and slices 0 .. C, tmp up to one slice past the
  // first chunk) go first, the rest behind them as a second outstanding exchange that travels while those launches run and is ended
  // in front of the first launch that reads it (the transport ends exchanges in the order they began).
  bool p_rest_pending = false;
  // Split exchanges begun and not yet ended.  The transports keep FIFO state per begin (comm_rccl.cpp: begun / ended and
  // the `arrived` events; TorchDistComm: its pending list), so an error return between a begin and its end must not leave
  // an entry behind -- the next exchange on this context would pop the stale one and read ghosts before they arrive.
  // Every error exit of this function therefore ends what it began (the context and its transport stay usable).
  struct OutstandingExchanges {
    bcg_context* c;
    int n = 0;
    ~OutstandingExchanges() {
      const std::string why = c->err;
      for (; n > 0; --n) (void)c->comm.halo_exchange_end(c->comm.user);
      c->err = why;
    }
  } outstanding{c};
  auto begin_window = [&](const double2* d, int lo, int n, int ring_slots, int b_lo, int b_n, int parity) -> int {
    BCG_TRY(halo_window(c, m, d, lo, n, ring_slots, /*split=*/true, b_lo, b_n, parity));
    if (c->distributed) outstanding.n += 1;
    return BCG_OK;
  };
  auto end_oldest = [&]() -> int {
    ProfScope ps(c, "halo_exchange_end");
    BCG_TRY(exchange_end(c));
    outstanding.n -= 1;
    return BCG_OK;
  };
  if (overlap && c->distributed) {
    const int n1 = (C + 1 < L3 - 1) ? C + 1 : L3 - 1;  // slices [0, n1) and slice L3 - 1
    BCG_TRY(begin_window(P->d, 0, n1, 0, L3 - 1, 1, par_p));
    if (n1 < L3 - 1) {
      BCG_TRY(begin_window(P->d, n1, L3 - 1 - n1, 0, 0, 0, par_p));
      p_rest_pending = true;
    }
    BCG_TRY(end_oldest());
  } else {
    BCG_TRY(halo_field(c, P));
  }
  // (a whole tmp keeps its slice 0: nothing is computed twice at the end of the sweep, no faces to put back)
  if (overlap && !half) BCG_TRY(slice0_faces(c, site_bytes, /*save=*/true));
  auto first = [&](int lo, int n) -> int {  // tmp[lo, lo+n) = D P
    if (half) {
      if (c->profiling) c->prof["stencil_form_k_hop4b_checkerboard"].count += 1;
    } else {
      note_stencil_form(c, m, 0, window(lo, n, 0), /*plain=*/true);
    }
    ProfScope ps(c, half ? "hop_half" : "hop_ring", alg_bytes(c, m, 2, kLinkPasses, n, vden), hop_flops(c, m, false, n, vden));
    const int nb = bcg::launch_hop_fast(c->stream, m, lat, g->U, g->Ughost, P->d, c->halo_recv, ring, bcg::HOP_PLAIN, nullptr,
                                        0.0, c->partials, false, kFastBlocks, tune, 0, window(lo, n, par_t));
    if (nb < 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "capacity mode: stencil window rejected");
    return check_launch(c, "hop_ring");
  };
  const double c0 = mass * mass + sigma0;
  int total = 0;
  auto second = [&](int lo, int hi) -> int {  // T[lo, hi) from tmp[lo - 1, hi]
    {
      if (half && c->profiling) c->prof["stencil_form_k_hop4b_checkerboard"].count += 1;
      ProfScope ps(c, half ? (gram ? "hop_half_shifted_gram" : "hop_half_shifted") : (gram ? "hop_shifted_gram_ring" : "hop_shifted_ring"),
                   alg_bytes(c, m, 3, kLinkPasses, hi - lo, vden), hop_flops(c, m, gram, hi - lo, vden));
      const int nb = bcg::launch_hop_fast(c->stream, m, lat, g->U, g->Ughost, ring, c->halo_recv, T->d, bcg::HOP_SHIFTED, P->d,
                                          c0, c->partials + static_cast<size_t>(total) * m * m, gram, kFastBlocks, tune, 0,
                                          window(lo, hi - lo, par_p));
      if (nb < 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "capacity mode: stencil window rejected");
      total += nb;
    }
    return check_launch(c, "hop_shifted_ring");
  };
  BCG_TRY(first(L3 - 1, 1));
  int next = 0;  // first slice of tmp not yet computed in order (L3 stands for slice 0 again)
  // tmp up to one slice past chunk [lo, hi); at the end of the sweep slice L3 = slice 0 again, from the source's faces of
  // slice 0: exchanged again (serial form) or put back from their copy (overlapped form: no second exchange in flight)
  auto stage_first = [&](int lo) -> int {
    const int hi = lo + C < L3 ? lo + C : L3;
    const int last = hi < L3 ? hi : L3 - 1;
    if (next <= last) BCG_TRY(first(next, last - next + 1));
    if (hi == L3 && !half) {
      if (overlap) BCG_TRY(slice0_faces(c, site_bytes, /*save=*/false));
      else BCG_TRY(halo_window(c, m, P->d, 0, 1, 0));  // its P faces were replaced by tmp faces of the first chunk
      BCG_TRY(first(0, 1));
    }
    next = hi + 1;
    return BCG_OK;
  };
  if (!overlap) {
    for (int lo = 0; lo < L3; lo += C) {
      const int hi = lo + C < L3 ? lo + C : L3;
      BCG_TRY(stage_first(lo));
      BCG_TRY(halo_window(c, m, ring, lo, hi - lo, half ? 0 : R, false, 0, 0, par_t));
      BCG_TRY(second(lo, hi));
    }
  } else {
    BCG_TRY(stage_first(0));
    BCG_TRY(begin_window(ring, 0, (C < L3 ? C : L3), half ? 0 : R, 0, 0, par_t));
    for (int lo = 0; lo < L3; lo += C) {
      const int hi = lo + C < L3 ? lo + C : L3;
      if (p_rest_pending) {  // the rest of the source's faces: posted before chunk 0's tmp faces, so ended before them
        BCG_TRY(end_oldest());
        p_rest_pending = false;
      }
      if (hi < L3) BCG_TRY(stage_first(hi));  // chunk k + 1's slices of tmp, while chunk k's faces are on the links
      BCG_TRY(end_oldest());
      if (hi < L3) BCG_TRY(begin_window(ring, hi, (hi + C < L3 ? C : L3 - hi), half ? 0 : R, 0, 0, par_t));
      BCG_TRY(second(lo, hi));  // ... and chunk k + 1's faces fly while chunk k's T is computed
    }
  }
  if (gram) *gram_blocks = total;
  return BCG_OK;
}

// The device memory apply_shifted needs for operands shaped like `like`, allocated now rather than at the first call
int reserve_operator_scratch(bcg_context* c, const bcg_field* like) {
  const int m = like->m;
  bcg_field* tmp;
  if (like->parity >= 0) {
    BCG_TRY(get_tmp_half(c, m, 1 - like->parity, &tmp));
    if (c->distributed) BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * 3 * m * sizeof(double2)));
    if (half_chunked_path(c) && fast_hop(c, m)) BCG_TRY(ensure_half_chunk_scratch(c, m));
    return BCG_OK;
  }
  if (fast_hop(c, m)) BCG_TRY(ensure_scratch(c));
  if (capacity_path(c, m)) return ensure_ring_scratch(c, m);
  BCG_TRY(get_tmp(c, m, &tmp));
  if (c->distributed) BCG_TRY(ensure_halo(c, static_cast<size_t>(c->ghost_sites) * 3 * m * sizeof(double2)));
  return BCG_OK;
}

// T = (mass^2 + sigma0) P - D(D(P))   [op + add(P, sigma0), inc/block_solvers.hpp:134-136]
int apply_shifted(bcg_context* c, const bcg_gauge* g, double mass, double sigma0, bcg_field* T, const bcg_field* P,
                  int* gram_blocks, bool* gram_folded) {
  if (gram_folded) *gram_folded = false;
  if (P->parity >= 0) {  // A restricted to one parity: tmp (other parity) = D P, T = (mass^2 + sigma0) P - D tmp
    if (gram_blocks) *gram_blocks = 0;
    if (T->parity != P->parity) BCG_FAIL(c, BCG_ERR_INVALID, "half-volume operator: result and argument must have the same parity");
    const int m = P->m;
    bcg_field* tmp;
    BCG_TRY(get_tmp_half(c, m, 1 - P->parity, &tmp));
    BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
    // the bundle sweep in its checkerboard form (m = 16, compact row a multiple of the tile, patch walk), else the generic kernel
    // (algorithmic bytes of these launches: two link passes per half site -- all the lattice's links, see apply_shifted_ring)
    bcg::LatticeDev latc = c->lat;
    latc.L[0] /= 2;
    latc.V /= 2;
    for (int mu = 1; mu < 4; ++mu) latc.stride[mu] /= 2;
    for (int mu = 0; mu < 4; ++mu) {  // half ghost faces: half the sites at half the offsets, compact in x0 like the field
      latc.face_sites[mu] /= 2;
      latc.ghost_off[mu][0] /= 2;
      latc.ghost_off[mu][1] /= 2;
    }
    // (direction 0 divided over ranks: the compact row's end sites would need the ghost face in one row parity only -- generic kernel)
    const bool fast = fast_hop(c, m) && (m == 16 || m == 32) && c->ndim == 4 && latc.L[0] > 0 && !c->lat.split[0] &&
                      bcg::hop_can_split_tiles(m, latc);
    // direction 3 whole, split exchange available: the sweep in x3 chunks with every exchange overlapped -- provided the
    // checkerboard bundle sweep takes EVERY window the chunked sweep launches (1, C and C + 1 slices and the last, shorter
    // chunk).  The chunked sweep has no generic fallback once its first exchange is posted; a tuning that switches the
    // bundle walk off (BCG_HOP_BUNDLE=0, an odd BCG_HOP_PATCH) or a slice too small for the grid lands in the blocking
    // path below, which falls back to k_hop_half.
    bool chunk_windows_ok = fast && half_chunked_path(c);
    if (chunk_windows_ok) {
      const int C = half_chunk(c), L3 = c->lat.L[3];
      for (int n = 1; n <= std::min(C + 1, L3) && chunk_windows_ok; ++n) {
        bcg::HopWindow w;
        w.x3_lo = 0;
        w.x3_n = n;
        w.cb = 1;
        chunk_windows_ok = bcg::hop_uses_bundle(m, latc, kFastBlocks, c->hop_tune, 0, w, /*plain=*/true);
      }
    }
    if (chunk_windows_ok) {
      int nb = 0;
      BCG_TRY(apply_shifted_ring(c, g, mass, sigma0, T, P, gram_blocks ? &nb : nullptr, tmp));
      if (gram_blocks) *gram_blocks = nb;
      return BCG_OK;
    }
    BCG_TRY(halo_field(c, P));  // (a lattice divided over ranks: the half faces of the source, then below those of tmp)
    int nb1 = -1, nb2 = -1;
    if (fast) {
      BCG_TRY(ensure_scratch(c));
      bcg::HopWindow w;
      w.cb = 1;
      w.cb_parity = tmp->parity;
      {
        ProfScope ps(c, "hop_half", alg_bytes(c, m, 2, 2, 1, 2), hop_flops(c, m, false, 1, 2));
        nb1 = bcg::launch_hop_fast(c->stream, m, latc, g->U, g->Ughost, P->d, c->halo_recv, tmp->d, bcg::HOP_PLAIN, nullptr, 0.0,
                                   c->partials, false, kFastBlocks, c->hop_tune, 0, w);
      }
      if (nb1 >= 0) {
        BCG_TRY(check_launch(c, "hop_half"));
        BCG_TRY(halo_field(c, tmp));
        const bool gram = gram_blocks != nullptr && m == 16;  // the fused product exists at m = 16 (as in the full-volume sweep)
        bcg::HopTuning tune = c->hop_tune;
        const bool fold = gram && gram_folded;
        if (fold) tune.fold = bcg::GramFold{c->dev_gram, c->fold_tickets};
        w.cb_parity = T->parity;
        {
          ProfScope ps(c, gram ? "hop_half_shifted_gram" : "hop_half_shifted", alg_bytes(c, m, 3, 2, 1, 2), hop_flops(c, m, gram, 1, 2));
          nb2 = bcg::launch_hop_fast(c->stream, m, latc, g->U, g->Ughost, tmp->d, c->halo_recv, T->d, bcg::HOP_SHIFTED, P->d,
                                     mass * mass + sigma0, c->partials, gram, kFastBlocks, tune, 0, w);
        }
        if (nb2 < 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "half-volume operator: second stencil rejected after the first ran");
        BCG_TRY(check_launch(c, "hop_half_shifted"));
        if (gram) {
          *gram_blocks = nb2;
          if (fold) *gram_folded = true;
        }
        if (c->profiling) c->prof["stencil_form_k_hop4b_checkerboard"].count += 2;
        return BCG_OK;
      }
    }
    {
      ProfScope ps(c, "hop_half", alg_bytes(c, m, 2, 2, 1, 2), hop_flops(c, m, false, 1, 2));
      bcg::launch_hop_half(c->stream, m, c->lat, tmp->parity, g->U, g->Ughost, P->d, c->halo_recv, tmp->d, bcg::HOP_PLAIN, nullptr, 0.0);
    }
    BCG_TRY(check_launch(c, "hop_half"));
    BCG_TRY(halo_field(c, tmp));
    {
      ProfScope ps(c, "hop_half_shifted", alg_bytes(c, m, 3, 2, 1, 2), hop_flops(c, m, false, 1, 2));
      bcg::launch_hop_half(c->stream, m, c->lat, T->parity, g->U, g->Ughost, tmp->d, c->halo_recv, T->d, bcg::HOP_SHIFTED, P->d,
                           mass * mass + sigma0);
    }
    return check_launch(c, "hop_half_shifted");
  }
  if (capacity_path(c, P->m)) return apply_shifted_ring(c, g, mass, sigma0, T, P, gram_blocks);
  bcg_field* tmp;
  BCG_TRY(get_tmp(c, P->m, &tmp));
  BCG_TRY(hop(c, g, tmp, P, bcg::HOP_PLAIN, nullptr, 0.0));
  return hop(c, g, T, tmp, bcg::HOP_SHIFTED, P, mass * mass + sigma0, gram_blocks, gram_folded);
}


}  // namespace bcg_impl

using namespace bcg_impl;

extern "C" {

int bcg_halo_plan(int ndim, const int* global_dims, const int* grid, const int* coords, size_t site_bytes, int* peer_send,
                  int* peer_recv, size_t* send_offset, size_t* recv_offset, size_t* nbytes, int64_t* ghost_sites) {
  if (ndim < 1 || ndim > 4 || !global_dims || !peer_send || !peer_recv || !send_offset || !recv_offset || !nbytes) return -1;
  return halo_plan(ndim, global_dims, grid, coords, site_bytes, peer_send, peer_recv, send_offset, recv_offset, nbytes,
                   ghost_sites);
}

// ---- operator ----------------------------------------------------------------------------------
int bcg_gauge_create(bcg_context* c, bcg_gauge** out) {
  DeviceScope on_device(c);
  if (!c || !out) return BCG_ERR_INVALID;
  bcg_gauge* g = new bcg_gauge{c, nullptr, nullptr, false};
  const size_t u_bytes = static_cast<size_t>(c->lat.V) * c->ndim * 9 * sizeof(double2);
  hipError_t e = hipMalloc(&g->U, u_bytes);
  if (e == hipSuccess && c->ghost_sites > 0) e = hipMalloc(&g->Ughost, static_cast<size_t>(c->ghost_sites) * 9 * sizeof(double2));
  if (e != hipSuccess) {
    if (g->U) (void)hipFree(g->U);
    delete g;
    c->err = std::string("bcg_gauge_create: hipMalloc: ") + hipGetErrorString(e);
    return BCG_ERR_HIP;
  }
  *out = g;
  return BCG_OK;
}

int bcg_gauge_destroy(bcg_gauge* g) {
  DeviceScope on_device(g ? g->ctx : nullptr);
  if (!g) return BCG_OK;
  (void)hipStreamSynchronize(g->ctx->stream);
  (void)hipFree(g->U);
  if (g->Ughost) (void)hipFree(g->Ughost);
  delete g;
  return BCG_OK;
}

int bcg_gauge_upload(bcg_gauge* g, const double* host) {
  DeviceScope on_device(g ? g->ctx : nullptr);
  if (!g || !host) return BCG_ERR_INVALID;
  bcg_context* c = g->ctx;
  HIP_TRY(c, hipMemcpyAsync(g->U, host, static_cast<size_t>(c->lat.V) * c->ndim * 9 * sizeof(double2),
                            hipMemcpyHostToDevice, c->stream));
  g->ghost_valid = false;
  return stream_sync(c);
}

int bcg_gauge_fill_random(bcg_gauge* g, uint64_t seed) {
  DeviceScope on_device(g ? g->ctx : nullptr);
  if (!g) return BCG_ERR_INVALID;
  bcg_context* c = g->ctx;
  bcg::launch_fill_gauge(c->stream, c->lat, c->gdims, g->U, seed);
  g->ghost_valid = false;
  return check_launch(c, "fill_gauge");
}

int bcg_dirac_hop(bcg_context* c, const bcg_gauge* g, bcg_field* out, const bcg_field* in) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(out, in) || out == in || g->ctx != c || in->ctx != c) return BCG_ERR_INVALID;
  return hop(c, g, out, in, bcg::HOP_PLAIN, nullptr, 0.0);
}

// out (parity p) = D in (parity 1 - p): the two off-diagonal blocks of D in the parity basis
int bcg_dirac_hop_half(bcg_context* c, const bcg_gauge* g, bcg_field* out, const bcg_field* in) {
  DeviceScope on_device(c);
  if (!c || !g || !out || !in || out == in || g->ctx != c || in->ctx != c || out->ctx != c || out->m != in->m || in->parity < 0 ||
      out->parity != 1 - in->parity)
    return BCG_ERR_INVALID;
  BCG_TRY(halo_gauge(c, const_cast<bcg_gauge*>(g)));
  BCG_TRY(halo_field(c, in));
  bcg::launch_hop_half(c->stream, in->m, c->lat, out->parity, g->U, g->Ughost, in->d, c->halo_recv, out->d, bcg::HOP_PLAIN, nullptr, 0.0);
  return check_launch(c, "hop_half");
}

int bcg_dirac_apply(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* out, const bcg_field* in) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(out, in) || out == in || g->ctx != c || in->ctx != c) return BCG_ERR_INVALID;
  return apply_shifted(c, g, mass, 0.0, out, in);
}

}  // extern "C"
