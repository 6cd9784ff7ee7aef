// nft_schemes.h -- host side: the splitting schemes of order 5..8 as "monomial programs".
//
// The reference lists the polynomial coefficients of these schemes one by one
// (src/private/fnft__akns_fscatter.c:435-912).  Here they are generated: every scheme is a
// Richardson-type combination  sum_n w_n Psi_n  of products of e^{aA} = diag(1, z^{a*deg}) and
// e^{bB} = expm([[0,q],[r,0]] b eps_t) = [[c_b, qs_b],[rs_b, c_b]]  (:46-59):
//   odd order p  (5A/B, 7A/B), n = 1,3,..,p : n alternating Lie-Trotter sub-steps
//        X(1/n) Y(2/n) X(2/n) ... X(2/n) Y(1/n)
//   even order p (6A/B, 8A/B), n = 1,..,p/2 : n Strang sub-steps
//        X(1/2n) [Y(1/n) X(1/n)]^(n-1) Y(1/n) X(1/2n)
//   with X = A for the "A" schemes, X = B for the "B" schemes, and
//   w_n = n^(2(T-1)) / prod_{m != n} (n^2 - m^2), T = number of terms.
// Multiplying a product out gives, for every matrix entry and power of z, a short sum of
// monomials  weight * prod(elements c_b / qs_b / rs_b); the device kernel (body_coeffs_prog)
// evaluates that list for every sample.  The list depends only on the scheme.
#pragma once
#include <algorithm>
#include <map>
#include <vector>

struct CoeffProgramHost {
    int deg = 0;
    int maxf = 0;                      // factors per monomial (padded with 255)
    std::vector<double> bfrac;         // distinct B step fractions of eps_t
    std::vector<int> tgt_ptr;          // CSR over targets t = e*(deg+1) + k (k: highest power first)
    std::vector<double> mw;            // weight of each monomial
    std::vector<unsigned char> mfac;   // maxf element ids per monomial: 3*j + {0: c, 1: qs, 2: rs}
};

namespace nft_schemes_detail {
struct Factor { bool is_B; int num, den; };

inline std::vector<Factor> sequence(bool odd, bool b_first, int n)
{
    std::vector<Factor> f;
    if (odd) {
        const int cnt = (n + 1) / 2;
        for (int i = 0; i < cnt; i++) {
            f.push_back({b_first, (i == 0) ? 1 : 2, n});
            f.push_back({!b_first, (i == cnt - 1) ? 1 : 2, n});
        }
    } else {
        f.push_back({b_first, 1, 2 * n});
        for (int i = 0; i < n; i++) {
            f.push_back({!b_first, 1, n});
            f.push_back({b_first, 1, (i == n - 1) ? 2 * n : n});
        }
    }
    return f;
}
struct Item { int row, col, zpow; std::vector<unsigned char> el; };
}  // namespace nft_schemes_detail

// akns_disc: ordinal of fnft__akns_discretization_t; returns false for schemes without a program
inline bool nft_build_coeff_program(int akns_disc, int deg, CoeffProgramHost &P)
{
    using namespace nft_schemes_detail;
    int order;
    bool b_first;
    switch (akns_disc) {
    case 11: order = 5; b_first = false; break;   // 2SPLIT5A
    case 12: order = 5; b_first = true; break;    // 2SPLIT5B
    case 13: order = 6; b_first = false; break;   // 2SPLIT6A
    case 14: order = 6; b_first = true; break;    // 2SPLIT6B
    case 15: order = 7; b_first = false; break;   // 2SPLIT7A
    case 16: order = 7; b_first = true; break;    // 2SPLIT7B
    case 17: order = 8; b_first = false; break;   // 2SPLIT8A
    case 18: order = 8; b_first = true; break;    // 2SPLIT8B
    default: return false;
    }
    const bool odd = order & 1;
    const int T = odd ? (order + 1) / 2 : order / 2;
    P = CoeffProgramHost();
    P.deg = deg;
    std::map<std::pair<int, int>, int> bindex;   // (num, den) reduced -> element block
    auto b_id = [&](int num, int den) {
        int a = num, b = den;
        while (b) { const int t = a % b; a = b; b = t; }
        const std::pair<int, int> key(num / a, den / a);
        auto it = bindex.find(key);
        if (it != bindex.end()) return it->second;
        const int id = (int)P.bfrac.size();
        bindex[key] = id;
        P.bfrac.push_back((double)key.first / (double)key.second);
        return id;
    };
    // (target, sorted elements) -> weight
    std::map<std::pair<int, std::vector<unsigned char>>, long double> mono;
    for (int t = 0; t < T; t++) {
        const int n = odd ? 2 * t + 1 : t + 1;
        long double w = 1.0L;
        for (int i = 0; i < 2 * (T - 1); i++) w *= (long double)n;
        for (int u = 0; u < T; u++) {
            const int m = odd ? 2 * u + 1 : u + 1;
            if (m != n) w /= (long double)(n * n - m * m);
        }
        std::vector<Item> items = {{0, 0, 0, {}}, {1, 1, 0, {}}};
        for (const Factor &f : sequence(odd, b_first, n)) {
            if (!f.is_B) {
                if ((f.num * deg) % f.den != 0) return false;
                const int k = f.num * deg / f.den;
                for (Item &it : items) if (it.col == 1) it.zpow += k;
            } else {
                const int j = b_id(f.num, f.den);
                std::vector<Item> next;
                for (const Item &it : items)
                    for (int c2 = 0; c2 < 2; c2++) {
                        Item nx = it;
                        nx.col = c2;
                        // [[c, qs],[rs, c]] element (it.col, c2)
                        const int comp = (it.col == c2) ? 0 : (it.col == 0 ? 1 : 2);
                        nx.el.push_back((unsigned char)(3 * j + comp));
                        next.push_back(nx);
                    }
                items.swap(next);
            }
        }
        for (Item &it : items) {
            if (it.zpow > deg) return false;
            std::sort(it.el.begin(), it.el.end());
            const int e = 2 * it.row + it.col;
            const int target = e * (deg + 1) + (deg - it.zpow);
            mono[{target, it.el}] += w;
        }
    }
    if (P.bfrac.size() > 8) return false;
    for (auto &kv : mono) P.maxf = std::max(P.maxf, (int)kv.first.second.size());
    P.tgt_ptr.assign((size_t)4 * (deg + 1) + 1, 0);
    for (auto &kv : mono) {   // map order = target-major, so the CSR fills in order
        P.tgt_ptr[(size_t)kv.first.first + 1]++;
        P.mw.push_back((double)kv.second);
        for (int f = 0; f < P.maxf; f++)
            P.mfac.push_back(f < (int)kv.first.second.size() ? kv.first.second[f] : (unsigned char)255);
    }
    for (size_t i = 1; i < P.tgt_ptr.size(); i++) P.tgt_ptr[i] += P.tgt_ptr[i - 1];
    return true;
}
