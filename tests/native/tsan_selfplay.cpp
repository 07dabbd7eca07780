#include <cstdio>
#include <initializer_list>
#include <cstring>
#include "../../include/cattus_selfplay.h"
int main() {
    for (int pass = 0; pass < 3; pass++) {
        const int game = pass == 0 ? CATTUS_GAME_HEX4 : CATTUS_GAME_CHESS;
        cattus_sp_config c; memset(&c, 0, sizeof c);
        c.struct_size = sizeof c; c.sim_num = game == CATTUS_GAME_CHESS ? 12 : 30; c.explore_factor = 1.41421f;
        c.temperature_count = 2; c.temperature_threshold[0] = 4; c.temperature_value[0] = 1.0f; c.temperature_threshold[1] = 9999; c.temperature_value[1] = 0.0f;
        c.prior_noise_alpha = 0.3f; c.prior_noise_epsilon = 0.25f;
        c.cache_size = 5000; c.batch_size = 8; c.threads = 6; c.concurrent_games = 24; c.seed = 3; c.game_stride = 1;
        c.leaves_in_flight = pass == 2 ? 4 : 1;  // last pass: several leaves per tree in flight
        uint32_t info[5]; cattus_sp_game_info(game, info);
        uint32_t ctx[2] = {info[1], info[2] * info[3]};
        cattus_sp_result* r = nullptr;
        int rc = cattus_sp_run(game, &c, cattus_sp_stub_net, ctx, nullptr, nullptr, 24, nullptr, nullptr, 1, &r);
        cattus_sp_summary s; cattus_sp_result_summary(r, &s);
        printf("game %d rc=%d w1=%u w2=%u d=%u positions=%llu evals=%llu\n", game, rc, s.player1_wins, s.player2_wins, s.draws,
               (unsigned long long)s.positions, (unsigned long long)s.node_evals);
        cattus_sp_result_free(r);
    }
}
