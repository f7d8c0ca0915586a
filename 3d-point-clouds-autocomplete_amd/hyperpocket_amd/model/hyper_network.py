"""Drop-in for model/hyper_network.py:5-43 (same names: model.{0,2,4,6,8}, output.{0..4})."""
import torch.nn as nn

from ..ops import HyperNetFunction


class HyperNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()

        self.input_size = config['input_size']
        self.use_bias = config['use_bias']
        self.relu_slope = config['relu_slope']
        if not self.use_bias:
            raise NotImplementedError("the HIP hypernetwork path expects use_bias=true")
        # target network layers out channels
        target_network_out_ch = [3] + config['target_network_layer_out_channels'] + [3]
        target_network_use_bias = int(config['target_network_use_bias'])

        self.model = nn.Sequential(
            nn.Linear(in_features=self.input_size, out_features=64, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Linear(in_features=64, out_features=128, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Linear(in_features=128, out_features=512, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Linear(in_features=512, out_features=1024, bias=self.use_bias),
            nn.ReLU(inplace=True),

            nn.Linear(in_features=1024, out_features=2048, bias=self.use_bias),
        )

        self.output = [
            nn.Linear(2048, (target_network_out_ch[x - 1] + target_network_use_bias) * target_network_out_ch[x],
                      bias=True)
            for x in range(1, len(target_network_out_ch))
        ]

        if not config['target_network_freeze_layers_learning']:
            self.output = nn.ModuleList(self.output)

    def forward(self, x):
        trunk = [self.model[i] for i in (0, 2, 4, 6, 8)]
        heads = list(self.output)
        params = [l.weight for l in trunk] + [l.bias for l in trunk] + [h.weight for h in heads] + [h.bias for h in heads]
        # (the engine that drives this step may take over the heads' weight gradient: ops.py)
        return HyperNetFunction.apply(x, len(heads), self.__dict__.get("_heads_exchange"), *params)
