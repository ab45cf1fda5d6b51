// ofdm_hip_tuning.h -- the laboratory keys of ofdm_set_tuning / ofdm_get_tuning (PRIVATE: tools/, tests/ and ofdm_abi.hip; not part of
// the drop-in contract of include/ofdm_hip.h, free to change between rounds).  One line per key: OFDM_TUNE_KEY(name, Tuning field,
// profile-build-only).  ofdm_abi.hip expands the list into its key table; value checks for keys that size LDS tiles or pick template
// instantiations are in ofdm_set_tuning.
//
//   A/B between kernel families (1 = take the older / generic kernel instead)
//     no_sc80             N = 64, W = 3 L Schmidl-Cox: the f32 filter pair k_sc_cf<128,first> + k_sc_cf<256,list> instead of k_sc80
//     no_sc_stream        L = 160 .. 5120: k_scb_chunks + k_scb_fine / k_sc_tile instead of the streaming detector k_sc_stream
//     no_sc_big           long periods through k_sc_tile instead of k_scb_chunks + k_scb_fine
//     no_fast64, no_demod4096, no_mid_kernels, no_rxframe1024, no_txframe64     the generic k_sym instead of that family
//     no_rx1024_finish    k_rxframe1024 writes raw bytes and k_rx_finish runs as its own launch
//     no_rxframe64_split  k_rxframe64 as one kernel with both frame bodies instead of the common-body / cut-body pair
//   shapes and depths
//     sc80_depth          k_sc80: steps between the last read of a ring piece and its refill (2: 7 KiB in flight per wavefront, 1: 9-10)
//     sc_wg_per_cu, sc_first_lags, sc128_one_wave                     k_sc_cf (the filter pair): workgroups per CU, lags of the first
//                         launch (0 = one launch), one wavefront per frame in the 128-chunk kernel
//     demod64_wg_per_cu, demod64_burst (16 / 8 / 4 / 1), demod64_narrow_stores,  k_demod64
//     demod64_store_policy (0, 1 nt, 2 sc1, 3 sc0 sc1)
//     tx_waves, txframe_keep_steps, txframe_rewrite,                  k_txframe64 / k_txframe_mid / k_txframe4096
//     no_txframe_optimistic
//     scb_two_segments, scb_big_tiles                                 k_scb_chunks / k_scb_fine
//   profile build only (libofdm_hip_profile.so): ablation exits and s_memtime section timers
//     debug_demod64, debug_sc, debug_tx
#ifndef OFDM_TUNE_KEY
#error "define OFDM_TUNE_KEY(name, field, profile_only) before including ofdm_hip_tuning.h"
#endif
OFDM_TUNE_KEY("no_sc80", no_sc80, false)
OFDM_TUNE_KEY("sc80_depth", sc80_depth, false)
OFDM_TUNE_KEY("no_sc_stream", no_sc_stream, false)
OFDM_TUNE_KEY("no_sc_big", no_sc_big, false)
OFDM_TUNE_KEY("no_fast64", no_fast64, false)
OFDM_TUNE_KEY("no_demod4096", no_demod4096, false)
OFDM_TUNE_KEY("no_mid_kernels", no_mid_kernels, false)
OFDM_TUNE_KEY("no_rxframe1024", no_rxframe1024, false)
OFDM_TUNE_KEY("no_txframe64", no_txframe64, false)
OFDM_TUNE_KEY("no_rx1024_finish", no_rx1024_finish, false)
OFDM_TUNE_KEY("no_rxframe64_split", no_rxframe64_split, false)
OFDM_TUNE_KEY("tx_waves", tx_waves, false)
OFDM_TUNE_KEY("txframe_keep_steps", txframe_keep_steps, false)
OFDM_TUNE_KEY("txframe_rewrite", txframe_rewrite, false)
OFDM_TUNE_KEY("no_txframe_optimistic", no_txframe_optimistic, false)
OFDM_TUNE_KEY("sc_wg_per_cu", sc_wg_per_cu, false)
OFDM_TUNE_KEY("sc_first_lags", sc_first_lags, false)
OFDM_TUNE_KEY("sc128_one_wave", sc128_one_wave, false)
OFDM_TUNE_KEY("demod64_wg_per_cu", demod64_wg_per_cu, false)
OFDM_TUNE_KEY("demod64_burst", demod64_burst, false)
OFDM_TUNE_KEY("demod64_narrow_stores", demod64_narrow_stores, false)
OFDM_TUNE_KEY("demod64_store_policy", demod64_store_policy, false)
OFDM_TUNE_KEY("scb_two_segments", scb_two_segments, false)
OFDM_TUNE_KEY("scb_big_tiles", scb_big_tiles, false)
OFDM_TUNE_KEY("debug_demod64", debug_demod64, true)
OFDM_TUNE_KEY("debug_sc", debug_sc, true)
OFDM_TUNE_KEY("debug_tx", debug_tx, true)
