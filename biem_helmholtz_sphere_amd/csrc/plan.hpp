// plan.hpp -- host + device tables of one (tree, n_end): labels, boundary-data quadrature, projection
// matrix and the translation terms of the closed form.  All k- and geometry-independent.
#pragma once
#include <vector>
#include <cstdint>
#include "special.hpp"

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
  // device mirrors (null until uploaded)
  int device = -1;
  int* d_labels = nullptr; int* d_deg = nullptr;
  int* d_labels2 = nullptr; int* d_deg2 = nullptr;
  int* d_units = nullptr;
  double* d_W = nullptr;
  uint32_t* d_ptr = nullptr; double* d_coef = nullptr; int32_t* d_tidx = nullptr;
  uint16_t* d_tidx16 = nullptr; int* d_chunk_ent = nullptr;
};

namespace biem {
int plan_build_host(biem_plan* p, int tree, int n_end);   // returns BIEM_* status
int plan_upload(biem_plan* p);
void plan_free(biem_plan* p);
void set_error(const char* fmt, ...);
}  // namespace biem
