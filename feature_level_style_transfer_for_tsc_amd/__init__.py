"""MI355X-native train step for the feature-level style-transfer TSC pipeline.

Drop-in module surface of BaeHann/feature_level_style_transfer_for_TSC's hot path (OS_CNN_res, OS_CNN,
WaveGlow, CPC, CDAN, …) whose convolutions, normalisations, flow steps and contrastive Gram products run
as hand-written gfx950 HIP kernels reached through the C ABI in include/fst_hip.h.  There is no CPU
fallback: using an op without libfst_hip.so or off an MI355X raises.
"""
from .structure import (generate_layer_parameter_list, get_Prime_number_in_a_range, get_out_channel_number,
                        layer_parameter_list_input_change, calculate_mask_index)
from .os_cnn import OS_CNN, OS_CNN_res, OS_block, Res_OS_layer, SampaddingConv1D_BN, build_layer_with_layer_parameter
from .waveglow import WaveGlow, WaveGlowLoss, WN, Invertible1x1Conv
from .cdan import CDAN, RandomLayer, Entropy
from .cpc import CPC
from .widgets import (AdversarialNetworkforCDAN, DimensionUnification, FeatureDiscriminatorforSource, NoiseTransfer,
                      ProbTransfer, wgan_loss)
from .step import ClassifierTrainer, JointConfig, JointTrainer, specs_for
from .dist import GradBucket, shard_batch
from .data import DeviceLoader, TestData, TrainData, load_ts, parse_ts
from .voting import collect_logits, multi_source_vote, multi_source_voting, precision_weights, vote_scores
from .checkpoint import (eval_accuracy, load_source_classification_modules, load_target_classification_modules,
                         save_source_classification_modules, save_target_classification_modules)

__version__ = "0.1.0"
