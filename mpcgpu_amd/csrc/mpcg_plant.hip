// mpcg_plant.hip — C ABI (include/mpcg.h) of the producer of the path's inputs: the robot as data (mpcg_plant) and the KKT block assembly
// mpcg_generate_kkt over the gfx950 kernel in kkt_plant.hip.h.
#include "mpcg_handle.hpp"
#include "kkt_plant.hip.h"

using namespace mpcg;

extern "C" {

// ---- the producer of the path's inputs: KKT block assembly with the robot as data (kkt_plant.hip.h) ----
struct mpcg_plant { int device = 0; PlantDev* d = nullptr; PlantDevT<float>* d32 = nullptr; };      // (d32: the same tables rounded to float, behind d in ONE allocation)

int mpcg_plant_create(mpcg_plant** out, int device, uint32_t num_joints, const double* X_const, const double* I_spatial, const double* Xhom_const,
                      const int32_t* X_trig_idx, const double* X_trig_coef, const int32_t* X_trig_j, uint32_t n_X_trig,
                      const int32_t* Xhom_trig_idx, const double* Xhom_trig_coef, const int32_t* Xhom_trig_j, uint32_t n_Xhom_trig) {
    if (!out) return MPCG_ERR_INVALID;
    *out = nullptr;
    if (num_joints != (uint32_t)PJ) return fail(nullptr, MPCG_ERR_UNSUPPORTED, "mpcg_plant_create: the compiled specialisation has 7 joints (IIWA-14)");
    if (!X_const || !I_spatial || !Xhom_const || (n_X_trig && (!X_trig_idx || !X_trig_coef || !X_trig_j)) ||
        (n_Xhom_trig && (!Xhom_trig_idx || !Xhom_trig_coef || !Xhom_trig_j)))
        return fail(nullptr, MPCG_ERR_INVALID, "mpcg_plant_create: null table");
    PlantDev* hp = new (std::nothrow) PlantDev();
    if (!hp) return MPCG_ERR_NOMEM;
    memset(hp, 0, sizeof(PlantDev));
    // The tables as given: X_k(q_k) = [[E, 0], [B, E]] with E = E0 + Es sin q_k + Ec cos q_k (likewise B), homogeneous transforms
    // R = R0 + Rs sin + Rc cos and translation p.  Tables are column-major (6x6 / 4x4), these are row-major 3x3 blocks.
    struct Given { double E0[PJ][9], Es[PJ][9], Ec[PJ][9], B0[PJ][9], Bs[PJ][9], Bc[PJ][9], R0[PJ][9], Rs[PJ][9], Rc[PJ][9], p[PJ][3], I[PJ][36]; };
    Given* gv = new (std::nothrow) Given();
    if (!gv) { delete hp; return MPCG_ERR_NOMEM; }
    memset(gv, 0, sizeof(Given));
    auto place = [&](int k, int r, int c, double v, int which /*0 const, 1 sin, 2 cos*/) -> bool {
        double(*E)[9] = which == 0 ? gv->E0 : (which == 1 ? gv->Es : gv->Ec);
        double(*B)[9] = which == 0 ? gv->B0 : (which == 1 ? gv->Bs : gv->Bc);
        if (r < 3 && c < 3) { E[k][3 * r + c] = v; return true; }
        if (r >= 3 && c < 3) { B[k][3 * (r - 3) + c] = v; return true; }
        return v == 0.0 || (r >= 3 && c >= 3);             // upper-right block must be zero; lower-right repeats E
    };
    bool ok = true;
    for (int k = 0; k < PJ; ++k) {
        for (int c = 0; c < 6; ++c)
            for (int r = 0; r < 6; ++r) {
                ok = ok && place(k, r, c, X_const[k * 36 + c * 6 + r], 0);
                gv->I[k][6 * r + c] = I_spatial[k * 36 + c * 6 + r];
            }
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) gv->R0[k][3 * r + c] = Xhom_const[k * 16 + c * 4 + r];
        for (int r = 0; r < 3; ++r) gv->p[k][r] = Xhom_const[k * 16 + 12 + r];
    }
    for (uint32_t t = 0; t < n_X_trig && ok; ++t) {
        const int idx = X_trig_idx[t], k = idx / 36, c = (idx % 36) / 6, r = idx % 6, j = X_trig_j[t];
        if (idx < 0 || k >= PJ || j < 0 || j >= 2 * PJ || j % PJ != k) { ok = false; break; }     // joint k's transform depends on q_k only
        // a trig entry REPLACES the constant at that position (load_update_XImats_helpers overwrites it)
        place(k, r, c, 0.0, 0);
        ok = place(k, r, c, X_trig_coef[t], j < PJ ? 1 : 2);
    }
    for (uint32_t t = 0; t < n_Xhom_trig && ok; ++t) {
        const int idx = Xhom_trig_idx[t], k = idx / 16, c = (idx % 16) / 4, r = idx % 4, j = Xhom_trig_j[t];
        if (idx < 0 || k >= PJ || j < 0 || j >= 2 * PJ || j % PJ != k || r >= 3 || c >= 3) { ok = false; break; }
        gv->R0[k][3 * r + c] = 0.0;
        (j < PJ ? gv->Rs : gv->Rc)[k][3 * r + c] = Xhom_trig_coef[t];
    }
    if (!ok) { delete hp; delete gv; return fail(nullptr, MPCG_ERR_INVALID, "mpcg_plant_create: tables do not describe a serial chain of revolute joints (X = [[E, 0], [B, E]], joint k depends on q_k)"); }
    // The kernel applies X_k(q) as blkdiag(Rz, Rz) Xtree, Rz = [[c, s, 0], [-s, c, 0], [0, 0, 1]] (a revolute joint about its own z axis,
    // the convention of GRiD's tables): row 0 = c T0 + s T1, row 1 = -s T0 + c T1, row 2 = T2 with T = E0 + Ec the transform at q = 0.
    // Verify that the given constant / sin / cos parts have exactly that form.
    double scale = 0.0;
    for (int k = 0; k < PJ; ++k)
        for (int e = 0; e < 9; ++e) {
            hp->ET[k][e] = gv->E0[k][e] + gv->Ec[k][e];
            hp->BT[k][e] = gv->B0[k][e] + gv->Bc[k][e];
            scale = fmax(scale, fmax(fabs(hp->ET[k][e]), fabs(hp->BT[k][e])));
        }
    auto rotz_form = [&](const double* T, const double* c0, const double* cs, const double* cc) {
        double worst = 0.0;
        for (int c = 0; c < 3; ++c) {
            worst = fmax(worst, fabs(cc[c] - T[c]) + fabs(cc[3 + c] - T[3 + c]) + fabs(cc[6 + c]));                 // cos part: rows 0, 1 of T
            worst = fmax(worst, fabs(cs[c] - T[3 + c]) + fabs(cs[3 + c] + T[c]) + fabs(cs[6 + c]));                 // sin part: T1, -T0
            worst = fmax(worst, fabs(c0[c]) + fabs(c0[3 + c]) + fabs(c0[6 + c] - T[6 + c]));                        // constant part: row 2
        }
        return worst;
    };
    double dev = 0.0;
    for (int k = 0; k < PJ; ++k) {
        dev = fmax(dev, rotz_form(hp->ET[k], gv->E0[k], gv->Es[k], gv->Ec[k]));
        dev = fmax(dev, rotz_form(hp->BT[k], gv->B0[k], gv->Bs[k], gv->Bc[k]));
    }
    if (!(dev <= 1e-12 * fmax(scale, 1.0))) {
        delete hp; delete gv;
        return fail(nullptr, MPCG_ERR_UNSUPPORTED, "mpcg_plant_create: every joint must rotate about its own z axis, X_k(q) = blkdiag(Rz(q), Rz(q)) X_k(0) (the form of GRiD's tables)");
    }
    // Spatial inertias: the kernel multiplies with the rigid-body form [[Ibar, skew(h)], [skew(h)^T, m 1]] (ten numbers).
    for (int k = 0; k < PJ; ++k) {
        const double* Ik = gv->I[k];
        double sc = 0.0, asym = 0.0;
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) { sc = fmax(sc, fabs(Ik[6 * r + c])); asym = fmax(asym, fabs(Ik[6 * r + c] - Ik[6 * c + r])); }
        if (!(asym <= 1e-12 * fmax(1.0, sc))) {
            delete hp; delete gv;
            return fail(nullptr, MPCG_ERR_INVALID, "mpcg_plant_create: spatial inertias must be symmetric");
        }
        const double mass = Ik[6 * 3 + 3], h[3] = {Ik[6 * 2 + 4], Ik[6 * 0 + 5], Ik[6 * 1 + 3]};      // skew(h) = [[0, -hz, hy], [hz, 0, -hx], [-hy, hx, 0]]
        const double sk[9] = {0, -h[2], h[1], h[2], 0, -h[0], -h[1], h[0], 0};
        double devI = 0.0;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                devI = fmax(devI, fabs(Ik[6 * r + 3 + c] - sk[3 * r + c]));
                devI = fmax(devI, fabs(Ik[6 * (3 + r) + 3 + c] - (r == c ? mass : 0.0)));
            }
        if (!(devI <= 1e-12 * fmax(1.0, sc))) {
            delete hp; delete gv;
            return fail(nullptr, MPCG_ERR_UNSUPPORTED, "mpcg_plant_create: spatial inertias must have the rigid-body form [[Ibar, skew(m c)], [skew(m c)^T, m 1]]");
        }
        const double ib[10] = {Ik[0], Ik[1], Ik[2], Ik[7], Ik[8], Ik[14], h[0], h[1], h[2], mass};
        memcpy(hp->Ib[k], ib, sizeof(ib));
    }
    // The end-effector position and Jacobian come out of the spatial transforms on the device (kkt_plant.hip.h, round 0); the reference
    // takes them from the homogeneous transforms.  Both tables describe the same chain: check it at three configurations.
    {
        const double qs[3][PJ] = {{0, 0, 0, 0, 0, 0, 0}, {0.3, -0.7, 1.1, 0.5, -1.3, 0.9, 0.2}, {-2.1, 1.4, -0.6, 1.9, 0.8, -1.7, 2.5}};
        double worst = 0.0;
        for (int t = 0; t < 3; ++t) {
            double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pos[3] = {0, 0, 0};
            double W[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, V[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};       // motion vectors [e_i; 0] pushed through the chain
            for (int k = 0; k < PJ; ++k) {
                const double sn = sin(qs[t][k]), cs = cos(qs[t][k]);
                double H[9], Rn[9];
                for (int e = 0; e < 9; ++e) H[e] = gv->R0[k][e] + gv->Rs[k][e] * sn + gv->Rc[k][e] * cs;
                for (int r = 0; r < 3; ++r) {
                    pos[r] += R[3 * r] * gv->p[k][0] + R[3 * r + 1] * gv->p[k][1] + R[3 * r + 2] * gv->p[k][2];
                    for (int c = 0; c < 3; ++c) Rn[3 * r + c] = R[3 * r] * H[c] + R[3 * r + 1] * H[3 + c] + R[3 * r + 2] * H[6 + c];
                }
                memcpy(R, Rn, sizeof(R));
                for (int i = 0; i < 3; ++i) {
                    double tw[3], tu[3];
                    for (int r = 0; r < 3; ++r) {
                        tw[r] = hp->ET[k][3 * r] * W[i][0] + hp->ET[k][3 * r + 1] * W[i][1] + hp->ET[k][3 * r + 2] * W[i][2];
                        tu[r] = hp->BT[k][3 * r] * W[i][0] + hp->BT[k][3 * r + 1] * W[i][1] + hp->BT[k][3 * r + 2] * W[i][2] +
                                hp->ET[k][3 * r] * V[i][0] + hp->ET[k][3 * r + 1] * V[i][1] + hp->ET[k][3 * r + 2] * V[i][2];
                    }
                    W[i][0] = cs * tw[0] + sn * tw[1]; W[i][1] = cs * tw[1] - sn * tw[0]; W[i][2] = tw[2];
                    V[i][0] = cs * tu[0] + sn * tu[1]; V[i][1] = cs * tu[1] - sn * tu[0]; V[i][2] = tu[2];
                }
            }
            const double ee[3] = {-(W[2][0] * V[1][0] + W[2][1] * V[1][1] + W[2][2] * V[1][2]), W[2][0] * V[0][0] + W[2][1] * V[0][1] + W[2][2] * V[0][2],
                                  -(W[1][0] * V[0][0] + W[1][1] * V[0][1] + W[1][2] * V[0][2])};
            for (int r = 0; r < 3; ++r) worst = fmax(worst, fabs(ee[r] - pos[r]));
        }
        if (!(worst <= 1e-9)) {
            delete hp; delete gv;
            return fail(nullptr, MPCG_ERR_INVALID, "mpcg_plant_create: the homogeneous transforms (Xhom) and the spatial transforms (X) describe different chains");
        }
    }
    delete gv;
    mpcg_plant* pl = new (std::nothrow) mpcg_plant();
    if (!pl) { delete hp; return MPCG_ERR_NOMEM; }
    if (device < 0 && hipGetDevice(&device) != hipSuccess) { delete hp; delete pl; return fail(nullptr, MPCG_ERR_HIP, "mpcg_plant_create: no HIP device"); }
    pl->device = device;
    // the float build of the kernel (linsys_t's own arithmetic, "kkt_f32") reads the same tables rounded to float
    PlantDevT<float>* hp32 = new (std::nothrow) PlantDevT<float>();
    if (!hp32) { delete hp; delete pl; return MPCG_ERR_NOMEM; }
    for (int k = 0; k < PJ; ++k) {
        for (int e = 0; e < 9; ++e) { hp32->ET[k][e] = (float)hp->ET[k][e]; hp32->BT[k][e] = (float)hp->BT[k][e]; }
        for (int e = 0; e < 10; ++e) hp32->Ib[k][e] = (float)hp->Ib[k][e];
    }
    if (hipSetDevice(device) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&pl->d), sizeof(PlantDev) + sizeof(PlantDevT<float>)) != hipSuccess ||
        hipMemcpy(pl->d, hp, sizeof(PlantDev), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(pl->d + 1, hp32, sizeof(PlantDevT<float>), hipMemcpyHostToDevice) != hipSuccess) {
        delete hp; delete hp32; delete pl;
        return fail(nullptr, MPCG_ERR_HIP, "mpcg_plant_create: cannot place the model on the device");
    }
    pl->d32 = reinterpret_cast<PlantDevT<float>*>(pl->d + 1);
    delete hp; delete hp32;
    *out = pl;
    return MPCG_OK;
}

// The KUKA LBR iiwa 14 the reference is built for, from the tables compiled into the library (csrc/iiwa14_model.inc): what
// gato_plant::initializeDynamicsConstMem<T>() returns in the reference (include/dynamics/iiwa/iiwa_eepos_plant.cuh:63-66).
#include "iiwa14_model.inc"
int mpcg_plant_create_iiwa14(mpcg_plant** out, int device) {
    return mpcg_plant_create(out, device, 7, kIiwa14_X_const, kIiwa14_I, kIiwa14_Xhom_const, kIiwa14_X_trig_idx, kIiwa14_X_trig_coef, kIiwa14_X_trig_j,
                             (uint32_t)(sizeof(kIiwa14_X_trig_idx) / sizeof(int32_t)), kIiwa14_Xhom_trig_idx, kIiwa14_Xhom_trig_coef, kIiwa14_Xhom_trig_j,
                             (uint32_t)(sizeof(kIiwa14_Xhom_trig_idx) / sizeof(int32_t)));
}

int mpcg_plant_destroy(mpcg_plant* p) {
    if (p && p->d) { RelaxedCaptureScope relaxed; (void)hipSetDevice(p->device); (void)hipFree(p->d); }
    delete p;
    return MPCG_OK;
}

int mpcg_generate_kkt(mpcg_handle* h, const mpcg_plant* plant, uint32_t control_size, float timestep, const float* d_eePos_traj,
                      const float* d_xs, const float* d_xu, float qd_cost, float r_cost, float* d_G_dense, float* d_C_dense,
                      float* d_g, float* d_c, uint32_t batch, void* stream) {
    if (!h || !plant) return MPCG_ERR_INVALID;
    if (!d_eePos_traj || !d_xs || !d_xu || !d_G_dense || !d_C_dense || !d_g || !d_c)
        return fail(h, MPCG_ERR_INVALID, "mpcg_generate_kkt: null device pointer");
    if (control_size != (uint32_t)PJ || h->n != 2u * PJ) return fail(h, MPCG_ERR_UNSUPPORTED, "mpcg_generate_kkt: state_size 14 / control_size 7 (IIWA-14) only");
    if (plant->device != h->device) return fail(h, MPCG_ERR_INVALID, "mpcg_generate_kkt: plant and handle live on different devices");
    if (batch == 0) return MPCG_OK;
    if (batch > h->max_batch) return fail(h, MPCG_ERR_INVALID, "mpcg_generate_kkt: batch exceeds max_batch");
    HIP_TRY(h, hipSetDevice(h->device));
    KktArgs a;
    a.plant = plant->d; a.eePos_traj = d_eePos_traj; a.xs = d_xs; a.xu = d_xu;
    a.G = d_G_dense; a.C = d_C_dense; a.g = d_g; a.c = d_c;
    a.N = (int)h->N; a.batch = (int)batch; a.dt = timestep; a.qd_cost = qd_cost; a.r_cost = r_cost;
    a.analytic = h->kkt_analytic;
    long blocks = ((long)batch * (h->N - 1) + KKT_ITEMS - 1) / KKT_ITEMS;      // one wavefront per KKT_ITEMS (trajectory, knot) pairs
    const long cap = (long)h->num_cus * 32;
    if (blocks > cap) blocks = cap;
    if (h->kkt_analytic && h->kkt_f32) {                  // linsys_t = float arithmetic throughout, as the reference's GRiD code (kkt_plant.hip.h, R = float)
        KktArgsT<float> f;
        f.plant = plant->d32; f.eePos_traj = d_eePos_traj; f.xs = d_xs; f.xu = d_xu;
        f.G = d_G_dense; f.C = d_C_dense; f.g = d_g; f.c = d_c;
        f.N = (int)h->N; f.batch = (int)batch; f.dt = timestep; f.qd_cost = qd_cost; f.r_cost = r_cost; f.analytic = 1;
        // 1: two knots per lane in packed float (whatever the size of the call: a trajectory's results do not depend on what else is in the batch);
        // 2: one knot per lane (the packed build's checker; 8 % faster than the default, where the packed build is 1.6x faster on throughput-sized calls)
        if (h->kkt_f32 == 1) {
            long pblocks = ((long)batch * (h->N - 1) + 2 * KKT_ITEMS - 1) / (2 * KKT_ITEMS);
            if (pblocks > cap) pblocks = cap;
            hipLaunchKernelGGL((generate_kkt_kernel<true, kkt_f2>), dim3((unsigned)pblocks), dim3(KKT_THREADS), 0, static_cast<hipStream_t>(stream), f);
        } else
        hipLaunchKernelGGL((generate_kkt_kernel<true, float>), dim3((unsigned)blocks), dim3(KKT_THREADS), 0, static_cast<hipStream_t>(stream), f);
    } else if (h->kkt_analytic) hipLaunchKernelGGL((generate_kkt_kernel<true, double>), dim3((unsigned)blocks), dim3(KKT_THREADS), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL((generate_kkt_kernel<false, double>), dim3((unsigned)blocks), dim3(KKT_THREADS), 0, static_cast<hipStream_t>(stream), a);
    HIP_TRY(h, hipGetLastError());
    return MPCG_OK;
}

}  // extern "C"
