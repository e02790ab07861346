"""The package's (few, optional) HECKA_* environment knobs, read ONCE at import - nothing on the call path reads the
environment.  `reload()` re-reads them and makes libhecka re-read its own (tests that switch a knob inside one process;
tests/conftest.py does it whenever a test changes a HECKA_* variable).  Every non-default value that survives here is
held to the default's results by a test (INTEGRATION.md 6 lists knob, default, test)."""
import os


class _Knobs(object):
    def read(self):
        env = os.environ.get

        def on(name, default='1'):
            return env(name, default) != '0'
        self.native16 = on('HECKA_NATIVE16')              # 16-bit features read natively (0: always the exact up-cast)
        self.ragged_x3 = env('HECKA_RAGGED_X3')           # None: by grid size; '1' / '0' force / forbid the masked f16x3 tile
        self.colmean_multi = on('HECKA_COLMEAN_MULTI')    # all layers' column means in one launch
        self.bwd_multi = on('HECKA_BWD_MULTI')            # all layers' dX in one launch
        self.bwd_tn = env('HECKA_BWD_TN', '256')          # column width of the all-layers dX tile (library knob; kernel label here)
        self.k32 = on('HECKA_K32')                        # 32-deep 16x16x32 tile (library knob; kernel label here)
        self.bwd_stream = on('HECKA_BWD_STREAM')          # streamed all-layers dX launch (library knob; kernel label here)
        self.multi_gram = on('HECKA_MULTI_GRAM')          # all layers' Grams in one launch
        self.fused_layers = on('HECKA_FUSED_LAYERS')      # one autograd node for all layers
        self.fused_ragged = on('HECKA_FUSED_RAGGED')      # ragged layers share the fused node
        self.conv_implicit = on('HECKA_CONV_IMPLICIT')    # implicit-GEMM convolutions
        self.conv_x3 = on('HECKA_CONV_X3')                # convolution / linear GEMMs on the fp16-split tile
        self.conv_direct = on('HECKA_CONV_DIRECT')        # few-channel direct convolution
        self.conv_tapmajor = on('HECKA_CONV_TAPMAJOR')    # implicit convolutions with K ordered (tap, channel) where 16 | C
        self.conv_keep_cols_gb = float(env('HECKA_CONV_KEEP_COLS_GB', '8'))
        self.bn_fuse = on('HECKA_BN_FUSE')                # norm + residual + relu in one launch
        self.bn_mask = on('HECKA_BN_MASK')                # the fused norm's backward reads a byte-per-quad relu mask instead of y
        self.sampler_torch = set(filter(None, env('HECKA_SAMPLER_TORCH', '').split(',')))
        self.shard_mode = env('HECKA_SHARD_MODE', 'slice')
        self.shard_exchange = env('HECKA_SHARD_EXCHANGE', 'a2a')
        self.shard_blocked = on('HECKA_SHARD_BLOCKED')    # one all_to_all_single per exchange on the blocked layouts
        self.shard_bwd_split = on('HECKA_SHARD_BWD_SPLIT')
        if self.shard_mode not in ('slice', 'gather'):
            raise ValueError("HECKA_SHARD_MODE must be 'slice' or 'gather'")
        if self.shard_exchange not in ('p2p', 'a2a'):
            raise ValueError("HECKA_SHARD_EXCHANGE must be 'p2p' or 'a2a'")
        return self


knobs = _Knobs().read()


def reload():
    knobs.read()
    from . import _lib
    if _lib._lib is not None:
        _lib.reload_config()
