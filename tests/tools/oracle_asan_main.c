/* Sanitizer driver for the CPU oracle (SURVEY §5 sanitizer row): exercises the emulator + wrapper
 * chain + frame pipeline + the scans under -fsanitize=address,undefined.  Built and run by
 * tests/test_oracle_sanitize.py (`make -C oracle asan`).  TEST INFRASTRUCTURE ONLY. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void* oracle_vec_new(const uint8_t* rom, uint32_t rom_size, int game, int E, int dim, uint64_t seed,
                     uint64_t env_id0, int64_t max_episode_steps);
void oracle_vec_free(void* p);
int oracle_vec_num_actions(void* p);
void oracle_vec_reset(void* p, uint8_t* obs);
void oracle_vec_step(void* p, const int64_t* actions, uint8_t* obs, float* rewards, uint8_t* dones);
int oracle_vtrace_f32(const float*, const float*, const float*, const float*, const float*, const float*,
                      float*, float*, int, int, float, float);
int oracle_gae_f32(const float*, const float*, const void*, const float*, const void*, float*, float*, int,
                   int, float, float, int, int, int);

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  static uint8_t rom[4096];
  uint32_t n = (uint32_t)fread(rom, 1, sizeof rom, f);
  fclose(f);
  const int game = atoi(argv[2]), steps = atoi(argv[3]);
  for (int dim = 42; dim <= 84; dim += 42) {
    enum { E = 3 };
    void* v = oracle_vec_new(rom, n, game, E, dim, 7, 0, 2000);
    const int A = oracle_vec_num_actions(v);
    uint8_t* obs = (uint8_t*)malloc((size_t)E * 4 * dim * dim);
    float rew[E];
    uint8_t done[E];
    int64_t act[E];
    oracle_vec_reset(v, obs);
    uint32_t s = 12345;
    long ndone = 0;
    for (int t = 0; t < steps; ++t) {
      for (int e = 0; e < E; ++e) { s = s * 1664525u + 1013904223u; act[e] = (s >> 16) % A; }
      oracle_vec_step(v, act, obs, rew, done);
      for (int e = 0; e < E; ++e) ndone += done[e];
    }
    printf("dim %d steps %d dones %ld obs0 %d\n", dim, steps, ndone, obs[0]);
    free(obs);
    oracle_vec_free(v);
  }
  enum { T = 49, B = 17 };
  float* x = (float*)malloc(sizeof(float) * T * B * 8);
  for (int i = 0; i < T * B * 8; ++i) x[i] = (float)((i * 37) % 101) / 101.0f - 0.5f;
  float boot[B];
  uint8_t d8[T * B];
  for (int i = 0; i < B; ++i) boot[i] = 0.25f * i;
  for (int i = 0; i < T * B; ++i) d8[i] = (i % 29) == 0;
  int rc = oracle_vtrace_f32(x, x + T * B, x + 2 * T * B, x + 3 * T * B, x + 4 * T * B, boot, x + 5 * T * B,
                             x + 6 * T * B, T, B, 1.0f, 1.0f);
  rc |= oracle_gae_f32(x, x + T * B, d8, boot, 0, x + 5 * T * B, x + 6 * T * B, T, B, 0.99f, 0.95f, 0, 0, 0);
  free(x);
  printf("scans rc %d\n", rc);
  return rc;
}
