"""Shared command-line flags (reference call surface: src/settings.py). Flag names, defaults and the
`Class__param` naming convention are the reference's, so its command lines run unchanged; flags whose
features are outside this build still parse (the factories raise when such a feature is selected)."""
from argparse import ArgumentParser, BooleanOptionalAction


class DefaultArgParser(ArgumentParser):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        flag = self.add_argument
        onoff = dict(action=BooleanOptionalAction)
        flag("--device", type=str, default="cpu")
        flag("--task", type=str)
        flag("--kernel", type=str, default=None)
        flag("--physics_true_adjoint", default=False, **onoff)
        flag("--sr_factor", type=int, default=None)
        flag("--noise_level", type=int, default=5)
        flag("--dataset", type=str, default="div2k")
        flag("--GroundTruthDataset__datasets_dir", type=str, default="./datasets")
        flag("--GroundTruthDataset__download", "--download", default=False, **onoff)
        size = self.add_mutually_exclusive_group()
        size.add_argument("--GroundTruthDataset__size", type=int, default=256)
        size.add_argument("--GroundTruthDataset__no_resize", action="store_const",
                          dest="GroundTruthDataset__size", const=None)
        flag("--SyntheticDataset__unique_seeds", default=True, **onoff)
        flag("--PrepareTrainingPairs__crop_size", type=int, default=256)
        flag("--PrepareTrainingPairs__crop_location", type=str, default="random")
        flag("--model_kind", type=str, default="Proposed")
        flag("--ProposedModel__architecture", type=str, default="Transformer")
        flag("--ConvolutionalModel__residual", default=True, **onoff)
        flag("--ConvolutionalModel__inner_residual", default=True, **onoff)
        flag("--ConvolutionalModel__inout_convs", default=True, **onoff)
        flag("--ConvolutionalModel__hidden_channels", type=int, default=32)
        flag("--ConvolutionalModel__scales", type=int, default=5)
        flag("--ConvolutionalModel__num_conv_blocks", type=int, default=1)
        flag("--SingleImageDataset__image_path", type=str, default=None)
        flag("--SingleImageDataset__duplicates_count", type=int, default=800)
        flag("--data_parallel_devices", type=str, default=None)
        flag("--physics_v2", default=True, **onoff)
