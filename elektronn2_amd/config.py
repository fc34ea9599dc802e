"""The handful of switches from elektronn2/config.py:57-96 that the hot path
consults.  (The cuDNN toggles use_manual_cudnn_* have no meaning here: there is
exactly one backend, libe2hip.so.)"""
use_ortho_init = False            # config.py: weight init mode for Conv
allow_floatX_downcast = True      # variables.py:148-155
time_per_step_smoothing_length = 50
loss_smoothing_length = 200
floatX = 'float32'
