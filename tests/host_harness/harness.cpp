// TEST-ONLY host build of df-vo_amd/csrc/solver_math.h (the per-lane device functions of the HIP
// solvers) so that their arithmetic can be checked bit-for-bit against the C oracle on a machine
// without a GPU.  Never linked into libdfvo_hip.so; the product has no CPU path.
#include "../../df-vo_amd/csrc/solver_math.h"

extern "C" {

int hh_five_point(const double* q1, const double* q2, double* E_out) {
    double EE[36], b[39], c[11], rre[10], rim[10];
    if (!sm::five_point_stage1(q1, q2, EE, b, c)) return 0;
    sm::solve_poly10(c, rre, rim);
    return sm::five_point_stage3(EE, b, rre, rim, E_out);
}
float hh_essential_error(const double* E, double a, double b, double c, double d) {
    return sm::essential_error(E, a, b, c, d);
}
int hh_homography_kernel(const float* M, const float* m, int count, double* H) {
    return sm::homography_kernel(M, m, count, H) ? 1 : 0;
}
int hh_homography_check_subset(const float* M, const float* m) { return sm::homography_check_subset(M, m) ? 1 : 0; }
float hh_homography_error(const double* H, float Mx, float My, float mx, float my) {
    float Hf[8];
    for (int i = 0; i < 8; i++) Hf[i] = (float)H[i];
    return sm::homography_error(Hf, Mx, My, mx, my);
}
void hh_decompose_essential(const double* E, double* R1, double* R2, double* t) { sm::decompose_essential(E, R1, R2, t); }
void hh_triangulate_point(const double* P1, const double* P2, double x1, double y1, double x2, double y2, double* X) {
    sm::triangulate_point(P1, P2, x1, y1, x2, y2, X);
}
int hh_ransac_update_num_iters(double p, double ep, int mp, int mi) { return sm::ransac_update_num_iters(p, ep, mp, mi); }
void hh_solve_eig8(const double* A, const double* b, double* x) { sm::solve_eig<8>(A, b, x); }
void hh_invert_eig8(const double* A, double* d) { sm::invert_eig<8>(A, d); }
}

// ---- numpy legacy RandomState + argpartition emulation
#include "../../df-vo_amd/csrc/kp_select.h"
#include "../../df-vo_amd/csrc/np_legacy.h"
extern "C" {
// state: key[624] + pos (as uint32[625]); functions update it in place
void hh_mt_shuffle(uint32_t* state, int n, int* perm) {
    sm::Mt19937 s;
    for (int i = 0; i < 624; i++) s.key[i] = state[i];
    s.pos = (int)state[624];
    sm::mt_shuffle_arange(s, n, perm);
    for (int i = 0; i < 624; i++) state[i] = s.key[i];
    state[624] = (uint32_t)s.pos;
}
void hh_mt_sample(uint32_t* state, int n_population, int n_samples, int* out) {
    sm::Mt19937 s;
    for (int i = 0; i < 624; i++) s.key[i] = state[i];
    s.pos = (int)state[624];
    int scratch[4096];
    sm::mt_sample_without_replacement(s, n_population, n_samples, out, scratch);
    for (int i = 0; i < 624; i++) state[i] = s.key[i];
    state[624] = (uint32_t)s.pos;
}
// keys-carried-along variant (what k_kp_cell runs out of LDS); padded so that the 4-wide scans may over-read
void hh_argpartition_cp(const float* v, int num, int kth, int* tosort) {
    float* key = new float[num + 8];
    for (int i = 0; i < num + 8; i++) key[i] = 0.f;
    for (int i = 0; i < num; i++) {
        tosort[i] = i;
        key[4 + i] = v[i];
    }
    if (num > 0) sm::kp_introselect_cp<int>(key + 4, tosort, num, kth, 0);
    delete[] key;
}
void hh_argpartition(const float* v, int num, int kth, int* tosort) {
    for (int i = 0; i < num; i++) tosort[i] = i;
    if (num > 0) sm::kp_introselect<int>(v, tosort, num, kth, 0);
}
void hh_cell_bounds(int h, int w, int nr, int nc, int row, int col, int* out) {
    sm::kp_cell_bounds(h, w, nr, nc, row, col, out, out + 1, out + 2, out + 3);
}
}
