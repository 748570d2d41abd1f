// Checker for the oracle's restatement of std::mt19937 + std::discrete_distribution: this program IS the C++ standard library
// (libstdc++, the one whisper.cpp links on Linux).  stdin: seed n_draws n, then n float weights; stdout: the draws.
#include <cstdio>
#include <random>
#include <vector>
int main() {
    unsigned seed; int n_draws, n;
    if (scanf("%u %d %d", &seed, &n_draws, &n) != 3) return 1;
    std::vector<float> w(n); for (int i = 0; i < n; ++i) if (scanf("%f", &w[i]) != 1) return 1;
    std::mt19937 rng(seed);
    for (int k = 0; k < n_draws; ++k) { std::discrete_distribution<> dist(w.begin(), w.end()); printf("%d\n", dist(rng)); }
    return 0;
}
