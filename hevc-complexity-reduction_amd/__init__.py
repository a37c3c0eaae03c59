"""hevc-complexity-reduction_amd -- MI355X-native ETH-CNN CU-partition predictor.

Host-side mirror (Python, as in the reference) of the one hot path this package replaces:
/root/reference/HM-16.5_Test_AI/bin/{video_to_cu_depth.py, net_CNN.py}, and of the LDP daemon
HM-16.5_Test_LDP/bin/resi_to_cu_depth_LDP.py (the "next" row: resi_cnn + one ETH-LSTM step).  All compute is in
lib/libethcnn.so (hand-written gfx950 kernels behind the C ABI of include/ethcnn.h).

The directory name has a hyphen; import it with
    importlib.import_module("hevc-complexity-reduction_amd")
"""
from . import ethcnn  # noqa: F401
from .ethcnn import EthCnn, EthCnnError, load_library  # noqa: F401
from . import net_CNN, sharding, video_to_cu_depth, resi_to_cu_depth_LDP  # noqa: F401
