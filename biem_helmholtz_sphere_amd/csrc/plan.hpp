// plan.hpp -- host + device tables of one (tree, n_end): labels, boundary-data quadrature, projection
// matrix and the translation terms of the closed form.  All k- and geometry-independent.
#pragma once
#include <vector>
#include <cstdint>
#include "special.hpp"

constexpr int kLists2dMax = 160;            // 2-D plans up to this order also carry term lists (tests, the generic fill forms)

struct biem_plan {
  int tree = 0, d = 0, n_end = 0, n2 = 0;   // n2 = 2 n_end - 1 (degrees of the translation table)
  int H = 0, H2 = 0, Q = 0;
  double Cd = 0.0;                          // (2 pi)^{d/2} sqrt(2/pi)
  // host tables
  std::vector<int> labels, deg;             // [H][3], [H]
  std::vector<int> labels2, deg2;           // [H2][3], [H2]
  std::vector<int> units;                   // [U][2]: harmonics (h, p) with conj Y_h = Y_p, h <= p (h == p: real harmonic)
  std::vector<double> qy, qw;               // [Q][d], [Q]
  std::vector<double> W;                    // [Q][H] complex128 interleaved
  std::vector<uint32_t> ptr;                // [H*H + 1]
  std::vector<double> coef;                 // [terms]
  std::vector<int32_t> tidx;                // [terms]
  std::vector<uint16_t> tidx16;             // same, 16-bit copy the fill kernel keeps in LDS
  std::vector<int> chunk_ent;               // fill chunks: entries e = h*H + h' in [chunk_ent[c], chunk_ent[c+1]); sized to the LDS budget
  int chunk_terms_max = 0;                  // largest number of terms in one chunk
  int chunk_ents_max = 0;                   // largest number of entries in one chunk
  bool lists_built = true;                  // false: 2-D beyond kLists2dMax - no term lists (ptr / coef / q* / r* are empty), only the 2-D fills run
  bool fill_table_global = false;           // general fill: the pair table stays in global memory (it does not fit LDS beside the term slice)
  // the symmetric (real-harmonic) fill: unit pairs (u, u') in row-major order, four term lists ("slots") per pair for the entries
  // (h,h'), (h,p'), (p,h'), (p,p') of the units (h,p), (h',p') - empty where h == p or h' == p' - so one thread forms a whole
  // 2 x 2 block of R W^H M W R^-1.  Internal order of the unknowns of a ball in that path: the U "cosine" combinations first
  // (slot u), then the "sine" combinations of the units with h != p (slot U + spos[u]); hpos[h] = slot holding harmonic h before
  // the transform (h -> u, p -> U + spos[u]).
  std::vector<int> spos, hpos;              // [U], [H]
  std::vector<uint32_t> qptr;               // [4 U U + 1]
  std::vector<double> qcoef;
  std::vector<uint16_t> qidx16;
  // entry-per-lane form: conjugation pairs the lists - (p,p') has the coefficients of (h,h') with every table index replaced by
  // its conjugate partner's, (p,h') those of (h,p') - so a unit pair carries TWO lists (A: (h,h'), B: (h,p'), B only when both units
  // are doubles) and the pair table is stored with partners adjacent: lin2[l] = 2e for the first member of unit e of the degree
  // < 2 n_end - 1 labels, 2e + 1 for its partner (a self-conjugate label fills both), so the partner's entry is index ^ 1.
  std::vector<int> lin2;                    // [H2]
  int H2lin = 0;                            // 2 * (number of units among the H2 labels)
  bool pair_lists_ok = false;               // the pairing was verified on every unit pair of this plan
  std::vector<uint32_t> q2ptr;              // [2 U U + 1]
  std::vector<double> q2coef;
  std::vector<uint16_t> q2idx16;            // indices into the paired table layout
  std::vector<int> qchunk;                  // chunks of unit pairs: [qchunk[c], qchunk[c+1]), at most FILL_SYM_THREADS pairs each
  int qchunk_terms_max = 0, qchunk_pairs_max = 0;
  // ---- reduced-table form of the entry-per-lane symmetric fill (k_fill_sym) ----
  // Every term of one entry (h, h') carries the same azimuthal order vector mu = m' - m, so T[l] = T'[e(l)] * e^{+-i mu.phi} with the
  // REAL-angular-factor table T'[e] = C_d h_{n''}(k|t|) |Y_l|-amplitude shared by a label and its conjugate partner (e = unit of
  // the label among the degree < 2 n_end - 1 labels) and the phase leaves the sum:  S(h,h') = phase * sum_p coef[p] T'[e[p]],  and
  // the conjugate entry (p,p') is conj(phase) times the SAME sum.  One chain per list instead of two, half the table.
  // Table row of a (pair, system): T'[0 .. E) then the NP phases e^{i mu_id . phi} of the distinct azimuthal vectors of first members.
  int E = 0, NP = 0;                        // E = H2lin / 2 label units, NP phase ids
  std::vector<int> red_of;                  // [H2] unit e of label l
  std::vector<int> red_first;               // [H2] 1: l is the first member of its unit (or self-conjugate), 0: the partner
  std::vector<int> red_label;               // [E] the first member's label index
  std::vector<int> ph_mu;                   // [NP][2] azimuthal orders of phase id (second entry: caa only)
  std::vector<int> ph_of_unit;              // [E] phase id of the unit's first member
  // term lists per wave of 64 consecutive unit pairs, TRANSPOSED and padded to the wave's longest list: row r holds term number
  // (r - start) of the 64 lanes (coefficient 0, index 0 where a lane's list has run out), so the per-step reads of a wave are
  // contiguous (no LDS bank conflicts).  Lists are padded to multiples of 4 rows; the indices of a group of 4 rows are packed per
  // lane (ridx[(group * 64 + lane) * 4 + k]: one 8-byte read per lane and group).  Chunk c = unit pairs [rchunk[c], rchunk[c+1]) (at most 1024 = 16 waves), its rows
  // [rcrow[c], rcrow[c+1]); rwrow[c * 33 + 2 w + {0, 1, 2}] = first row of wave w's list A, of its list B, end (relative to the chunk).
  bool red_lists_ok = false;
  int red_waves = 16, red_nc = 2;           // workgroup of k_fill_red (waves) and combinations per iteration the chunks were cut for (experiments: BIEM_FILL_RED_WAVES=8, BIEM_FILL_NC=1 at plan build)
  std::vector<double> rcoef;                // [rows][64]
  std::vector<uint16_t> ridx;               // [rows][64] indices e into T'
  std::vector<uint16_t> rphsel;             // [U U][2]: phase selector of list A, of list B: 2 * phase id + (1: conjugate phase)
  std::vector<int> rchunk, rcrow, rwrow;
  int rchunk_rows_max = 0;
  double red_gather_cycles = 0.0;           // LDS cycles per row and ds_read_b128 lane group of the table gather (1 = conflict-free)
  // the same lists cut into small chunks for the systems-in-lanes form of the symmetric fill (k_fill_sys: no LDS ceiling on H2)
  std::vector<int> schunk;
  int schunk_terms_max = 0, schunk_pairs_max = 0;
  // device mirrors (null until uploaded)
  int device = -1;
  int* d_labels = nullptr; int* d_deg = nullptr;
  int* d_labels2 = nullptr; int* d_deg2 = nullptr;
  int* d_units = nullptr;
  double* d_W = nullptr;
  uint32_t* d_ptr = nullptr; double* d_coef = nullptr; int32_t* d_tidx = nullptr;
  uint16_t* d_tidx16 = nullptr; int* d_chunk_ent = nullptr;
  int* d_spos = nullptr; int* d_hpos = nullptr; uint32_t* d_qptr = nullptr; double* d_qcoef = nullptr; uint16_t* d_qidx16 = nullptr;
  int* d_qchunk = nullptr; int* d_schunk = nullptr;
  int* d_lin2 = nullptr; uint32_t* d_q2ptr = nullptr; double* d_q2coef = nullptr; uint16_t* d_q2idx16 = nullptr;
  int* d_red_of = nullptr; int* d_red_first = nullptr; int* d_red_label = nullptr; int* d_ph_mu = nullptr;
  double* d_rcoef = nullptr; uint16_t* d_ridx = nullptr; uint16_t* d_rphsel = nullptr; int* d_rchunk = nullptr; int* d_rcrow = nullptr; int* d_rwrow = nullptr;
};

namespace biem {
int plan_build_host(biem_plan* p, int tree, int n_end);   // returns BIEM_* status
int plan_upload(biem_plan* p);
void plan_free(biem_plan* p);
void set_error(const char* fmt, ...);
}  // namespace biem
