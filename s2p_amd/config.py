"""s2p_amd/config.py -- the subset of the reference's global ``cfg`` dict that the hot path reads
(s2p/config.py:12-179).  When the reference package is importable its own dict object is used, so a
config.json loaded by s2p.read_config_file reaches the HIP path unchanged; otherwise the same
defaults are provided here."""

try:                                     # the untouched orchestrator's dict, if s2p is installed
    from s2p.config import cfg           # noqa: F401
except Exception:                        # standalone use (tests, bench): same defaults
    cfg = {}
    cfg['temporary_dir'] = '/tmp'                       # s2p/config.py:18 ($TMPDIR there)
    cfg['clean_tmp'] = True                             # :21
    cfg['omp_num_threads'] = 1                          # :46
    cfg['timeout'] = 600                                # :50
    cfg['max_disp_range'] = None                        # :77
    cfg['matching_algorithm'] = 'mgm'                   # :136
    cfg['census_ncc_win'] = 5                           # :139
    cfg['stereo_speckle_filter'] = 25                   # :142
    cfg['stereo_regularity_multiplier'] = 1.0           # :145
    cfg['mgm_nb_directions'] = 8                        # :149
    cfg['mgm_timeout'] = 600                            # :151
    cfg['mgm_leftright_threshold'] = 1.0                # :153
    cfg['mgm_leftright_control'] = 1                    # :157
    cfg['mgm_mindiff_control'] = -1                     # :160
    cfg['horizontal_margin'] = 50                       # :35
    cfg['vertical_margin'] = 10                         # :36
    cfg['msk_erosion'] = 2                              # :107
