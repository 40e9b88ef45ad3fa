// host_driver.cpp -- the solve-once / parameter-scan driver above the C ABI, and the null
// vector of the converged matrix.
//
// Mirrors reference src/main.cpp: solve_once_eigen (:19-80), the {head, step, tail} scan
// generator with its two-direction sweep and 0.01*step fuzz (:139-172), filter_input
// (:174-180), the scan loop with omega continuation and the NaN fall-back (:223-325), the
// output.json schema (:208-221, 246-262, 270-321) and the raw eigenMatrics/<key>Eq<v>.bin files
// (:255-257, 295-299, 61-63).  nullSpace (include/solver.h:58-112) runs on the device
// (emme_null_vectors_batch, nullspace.hip); the host version below (inverse iteration on M^H M with a
// partial-pivot LU of a complex SYMMETRIC M) is emme_null_vector: a second opinion for tests and tools.
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/emme_hip.h"
#include "host_json.hpp"

namespace emme {
void set_error(const std::string& msg);
int params_from_value(const JsonValue& root, emme_params_t* out);
}

namespace {

using emme::JsonValue;
using cplx = std::complex<double>;

// ---- null vector: the right singular vector of the smallest singular value, by inverse iteration -------
// (M^H M)^-1 is applied as M^-1 M^-H through ONE LU of M itself (M complex symmetric => M^-H v =
// conj(M^-1 conj(v))): M^H M is never formed, so the factorisation sees cond(M), not its square; the
// convergence factor per sweep is (s_n / s_n-1)^2, four sweeps.
struct Lu {
    int n;
    std::vector<cplx> a;
    std::vector<int> piv;
    bool factor(const cplx* m, int n_) {
        n = n_;
        a.assign(m, m + (size_t)n * n);
        piv.resize(n);
        for (int k = 0; k < n; ++k) {
            int p = k;
            double best = std::abs(a[(size_t)k * n + k]);
            for (int r = k + 1; r < n; ++r) {
                const double v = std::abs(a[(size_t)r * n + k]);
                if (v > best) best = v, p = r;
            }
            piv[k] = p;
            if (p != k)
                for (int c = 0; c < n; ++c) std::swap(a[(size_t)k * n + c], a[(size_t)p * n + c]);
            cplx d = a[(size_t)k * n + k];
            if (d == cplx(0.0)) d = a[(size_t)k * n + k] = cplx(1e-300);  // exactly singular: perturb
            for (int r = k + 1; r < n; ++r) {
                const cplx f = a[(size_t)r * n + k] / d;
                a[(size_t)r * n + k] = f;
                if (f == cplx(0.0)) continue;
                for (int c = k + 1; c < n; ++c) a[(size_t)r * n + c] -= f * a[(size_t)k * n + c];
            }
        }
        return true;
    }
    void solve(std::vector<cplx>& x) const {
        for (int k = 0; k < n; ++k) {
            std::swap(x[k], x[piv[k]]);
            for (int r = k + 1; r < n; ++r) x[r] -= a[(size_t)r * n + k] * x[k];
        }
        for (int r = n - 1; r >= 0; --r) {
            cplx s = x[r];
            for (int c = r + 1; c < n; ++c) s -= a[(size_t)r * n + c] * x[c];
            x[r] = s / a[(size_t)r * n + r];
        }
    }
};

void null_vector_of(const cplx* m, int n, cplx* out) {
    Lu lu;
    lu.factor(m, n);
    std::vector<cplx> v(n);
    for (int i = 0; i < n; ++i) v[i] = cplx(1.0 + 0.37 * std::sin(1.0 + i), 0.21 * std::cos(2.0 * i));
    auto normalise = [&](std::vector<cplx>& x) {
        double s = 0;
        for (auto& e : x) s += std::norm(e);
        s = 1.0 / std::sqrt(s);
        for (auto& e : x) e *= s;
    };
    normalise(v);
    for (int it = 0; it < 4; ++it) {
        // u = M^-H v = conj(M^-1 conj(v));  v = M^-1 u
        for (auto& e : v) e = std::conj(e);
        lu.solve(v);
        for (auto& e : v) e = std::conj(e);
        normalise(v);
        lu.solve(v);
        normalise(v);
    }
    for (int i = 0; i < n; ++i) out[i] = v[i];
}

// ---- driver ---------------------------------------------------------------------------------
std::string date_string() {
    // reference src/functions.cpp:8-20: %FT%T%z with a colon inserted in the zone offset
    std::time_t t = std::time(nullptr);
    std::tm tm = *std::localtime(&t);
    char buf[64];
    std::strftime(buf, sizeof buf, "%FT%T%z", &tm);
    std::string s(buf);
    if (s.size() >= 5) {
        const char c = s[s.size() - 5];
        if (c == '+' || c == '-') s.insert(s.size() - 2, ":");
    }
    return s;
}

JsonValue filter_input(const JsonValue& all) {
    JsonValue in = all;
    for (auto& m : in.members)
        if (m.second.is_object()) m.second = JsonValue(m.second.at("head"));
    return in;
}

// get_scan_generator, src/main.cpp:139-172
struct ScanGen {
    double head, step, left_tail, right_tail, current, current_tail;
    bool to_left = true, is_first = true;
    ScanGen(double h, double s, double l, double r)
        : head(h), step(s), left_tail(l), right_tail(r), current(h), current_tail(l) {}
    bool within() const {
        return std::abs(current - head) <= (std::abs(current_tail - head) + 0.01 * std::abs(step));
    }
    // returns (continue, turning, value)
    void next(bool& cont, bool& turning, double& value) {
        if (!is_first) current += std::copysign(step, (current_tail - head));
        is_first = false;
        if (within()) {
            cont = true, turning = false, value = current;
            return;
        }
        to_left = !to_left;
        current_tail = right_tail;
        current = head + std::copysign(step, (current_tail - head));
        cont = !to_left && within();
        turning = true;
        value = current;
    }
};

// solve_once_eigen on an input object whose scan keys have been replaced by scalars
JsonValue solve_once(const JsonValue& input, cplx& omega, const std::string& matrix_path, bool& file_ok) {
    emme_params_t p;
    if (emme::params_from_value(input, &p) != EMME_OK) throw std::runtime_error(emme_last_error());
    emme_ctx_t* ctx = nullptr;
    if (emme_ctx_create(&p, -1, &ctx) != EMME_OK) throw std::runtime_error(emme_last_error());
    struct Guard {
        emme_ctx_t* c;
        ~Guard() { emme_ctx_destroy(c); }
    } guard{ctx};
    const double g[2] = {omega.real(), omega.imag()};
    double root[2];
    int iters = 0, info = 0;
    if (emme_solve_roots(ctx, g, 1, p.iteration_precision, p.iteration_step_limit, root, &iters, &info,
                         nullptr) != EMME_OK)
        throw std::runtime_error(emme_last_error());
    if (info > 0) {
        // include/solver.h:142-153.  The reference prints zsysv's index of the singular block of its LDL^T
        // factorisation; the step here is a partial-pivot LU, whose index of the zero pivot is a different
        // number for the same verdict: the reference's sentence without a number it would not print, and
        // this library's index marked as such.
        std::ostringstream oss;
        oss << "Linear solve failed. The factorization has been completed, but the "
            << "block diagonal matrix D is exactly singular"
            << ", so the solution could not be computed. [emme_amd: partial-pivot LU, zero pivot in column " << info << "]";
        throw std::runtime_error(oss.str());
    }
    if (info < 0) {
        // Failures the reference does not have a code for.  It would carry a non-finite integral
        // into M, where zsysv fails (the message above), or into omega and end with a NaN eigenvalue;
        // either way the scan records {"eigenvalue":"NaN","reason":...} (src/main.cpp:311-318) and does NOT
        // continue the next scan point from this omega.  Same here.
        throw std::runtime_error(
            info == EMME_ENUMERIC
                ? "Linear solve failed. The matrix holds a non-finite integral, or the quadrature depth cap was reached "
                  "(EMME_ENUMERIC), so the solution could not be computed."
                : "Linear solve failed on the device (EMME_EDEVICE).");
    }
    const int dim = emme_ctx_dim(ctx);
    std::vector<cplx> M((size_t)dim * dim), vec(dim);
    if (emme_ctx_get_matrix(ctx, 0, reinterpret_cast<double*>(M.data())) != EMME_OK)
        throw std::runtime_error(emme_last_error());
    file_ok = false;
    if (!matrix_path.empty()) {
        std::ofstream f(matrix_path, std::ios::binary);  // src/main.cpp:61-63
        f.write(reinterpret_cast<const char*>(M.data()), (std::streamsize)(sizeof(cplx) * M.size()));
        file_ok = (bool)f;
    }
    // nullSpace (include/solver.h:58-112, src/main.cpp:73-76) on the device, from M(omega_final) as the search left it
    int vinfo = 0;
    if (emme_null_vectors_batch(ctx, dim, 1, nullptr, reinterpret_cast<double*>(vec.data()), &vinfo) != EMME_OK)
        throw std::runtime_error(emme_last_error());
    JsonValue res = JsonValue::make_object();
    JsonValue ev = JsonValue::make_array();
    ev.items.push_back(JsonValue::make(root[0]));
    ev.items.push_back(JsonValue::make(root[1]));
    res["eigenvalue"] = ev;
    JsonValue evec = JsonValue::make_array();
    for (int i = 0; i < dim; ++i) {
        JsonValue pr = JsonValue::make_array();
        pr.items.push_back(JsonValue::make(vec[i].real()));
        pr.items.push_back(JsonValue::make(vec[i].imag()));
        evec.items.push_back(pr);
    }
    res["eigenvector"] = evec;
    res["iterations"] = JsonValue::make(iters);
    omega = cplx(root[0], root[1]);  // continuation, src/main.cpp:78
    return res;
}

}  // namespace

extern "C" {

int emme_null_vector(const double* M, int n, double* vec) {
    if (!M || !vec || n < 1) return EMME_EINVAL;
    null_vector_of(reinterpret_cast<const cplx*>(M), n, reinterpret_cast<cplx*>(vec));
    return EMME_OK;
}

void emme_free(void* p) { std::free(p); }

int emme_scan_values(double head, double step, double tail0, double tail1, double* values,
                     int* turning_flags, int max_values) {
    // the value sequence the scan loop of src/main.cpp:264-324 visits for one axis
    ScanGen gen(head, step, tail0, tail1);
    bool cont, turning;
    double v;
    int n = 0;
    gen.next(cont, turning, v);
    while (cont && n < max_values) {
        if (values) values[n] = v;
        if (turning_flags) turning_flags[n] = turning ? 1 : 0;
        ++n;
        gen.next(cont, turning, v);
    }
    return n;
}

int emme_run_json(const char* input_text, const char* matrix_dir, char** output_text) {
    if (!input_text || !output_text) return EMME_EINVAL;
    *output_text = nullptr;
    try {
        const JsonValue all = emme::json_parse(input_text, "input.json");
        const std::string method = all.at("method").str();
        if (method != "eigen") {
            std::ostringstream oss;
            oss << "Method '" << method << "' is not supported, yet.\n";  // src/main.cpp:193-196
            throw std::runtime_error(oss.str());
        }
        const JsonValue& guess = all.at("initial_guess");
        const cplx guess0(guess.at((size_t)0).number(), guess.at((size_t)1).number());
        const std::string dir = matrix_dir ? std::string(matrix_dir) : std::string();

        JsonValue result = JsonValue::make_object();
        result["input"] = all;
        result["run_time"] = JsonValue::make(date_string());
        JsonValue results = JsonValue::make_object();

        // scan axes: every top-level key whose value is an object (src/main.cpp:225-242)
        struct Axis {
            std::string key;
            double head, step, t0, t1;
            bool head_is_int;
        };
        std::vector<Axis> axes;
        for (const auto& m : all.members) {
            if (!m.second.is_object()) continue;
            Axis a;
            a.key = m.first;
            a.head = m.second.at("head").number();
            a.head_is_int = m.second.at("head").kind == JsonValue::Int;
            a.step = m.second.at("step").number();
            const JsonValue& tail = m.second.at("tail");
            if (tail.is_array()) {
                a.t0 = tail.at((size_t)0).number();
                a.t1 = tail.at((size_t)1).number();
            } else {
                a.t0 = tail.number();
                a.t1 = a.head + .5 * std::copysign(a.step, a.head - a.t0);
            }
            axes.push_back(a);
        }

        if (axes.empty()) {
            JsonValue unit = JsonValue::make_object();
            unit["scan_key"] = JsonValue::make(std::string("(None)"));
            JsonValue arr = JsonValue::make_array();
            cplx omega = guess0;
            bool ok = false;
            arr.items.push_back(solve_once(all, omega, dir.empty() ? "" : dir + "/eigenMatrix.bin", ok));
            unit["scan_result"] = arr;
            results["(None)"] = unit;
        } else {
            for (const Axis& ax : axes) {
                JsonValue input = filter_input(all);
                ScanGen gen(ax.head, ax.step, ax.t0, ax.t1);
                bool cont, turning;
                double value;
                gen.next(cont, turning, value);
                JsonValue unit = JsonValue::make_object();
                unit["scan_key"] = JsonValue::make(ax.key);
                JsonValue values = JsonValue::make_array();
                JsonValue arr = JsonValue::make_array();
                cplx omega = guess0;
                while (cont) {
                    // Value::operator=(double) keeps the category of the head value: an
                    // INTEGER head truncates every scan value (include/JsonParser.h:176-186)
                    input[ax.key] = ax.head_is_int ? JsonValue::make((int)value) : JsonValue::make(value);
                    values.items.push_back(JsonValue::make(value));
                    if (turning) {  // second direction restarts from the head's root (:282-291)
                        const JsonValue& first = arr.items.at(0).at("eigenvalue");
                        if (first.is_string())
                            omega = guess0;
                        else
                            omega = cplx(first.at((size_t)0).number(), first.at((size_t)1).number());
                    }
                    const std::string fname =
                        dir.empty() ? "" : dir + "/" + ax.key + "Eq" + std::to_string(value) + ".bin";
                    try {
                        bool ok = false;
                        JsonValue one = solve_once(input, omega, fname, ok);
                        one["eigenMatrix"] = JsonValue::make(ok ? fname : "Can not open '" + fname + "' for write.");
                        one["scan_value"] = JsonValue::make(value);
                        arr.items.push_back(one);
                    } catch (const std::exception& e) {
                        JsonValue err = JsonValue::make_object();
                        err["eigenvalue"] = JsonValue::make(std::string("NaN"));
                        err["reason"] = JsonValue::make(std::string(e.what()));
                        arr.items.push_back(err);
                    }
                    gen.next(cont, turning, value);
                }
                unit["scan_values"] = values;
                unit["scan_result"] = arr;
                results[ax.key] = unit;
            }
        }
        result["result"] = results;
        const std::string text = result.dump(0);
        *output_text = static_cast<char*>(std::malloc(text.size() + 1));
        if (!*output_text) return EMME_ENOMEM;
        std::memcpy(*output_text, text.c_str(), text.size() + 1);
        return EMME_OK;
    } catch (const std::exception& e) {
        emme::set_error(e.what());
        return EMME_EJSON;
    }
}

}  // extern "C"
