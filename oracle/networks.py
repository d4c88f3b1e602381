"""Oracle restatement of the reference networks (TEST INFRASTRUCTURE ONLY, CPU, literal per-edge
formulation).  Follows /root/reference/Model/models.py line by line in behaviour:

  position table ............ models.py:13-19      floor level .............. models.py:22-25
  MLP builder ............... models.py:29-42      pos / non-pos split ...... models.py:45-49
  ProgramBlock / ProgramGNN . models.py:52-92      Attention ................ models.py:95-141
  VoxelBlock ................ models.py:144-172    VoxelGNN_G / VoxelGNN_D .. models.py:175-229
  Generator ................. models.py:232-250    DiscriminatorVoxel ....... models.py:253-264

Parameter / buffer names equal the reference's, so state_dicts (and runs/iccv2021/checkpoints/
70.ckpt with strict=False) load unchanged.  Gumbel / Exp(1) noise is an explicit argument:
``noise`` is a sequence with one ``(exp_mask[Nv,2], exp_att[Ec])`` pair per Attention call, in call
order (6 per generator forward at voxel_layer=12); None draws the noise like the reference.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .pointer import (dense_gumbel_softmax, find_max_out_program_index, segment_gumbel_softmax,
                      straight_through_one_hot)
from .scatter import (avg_pool_x, global_max_pool, glorot_, max_pool_x, propagate_mean, scatter_sum)

NUMBER_OF_CLASS = 6  # reference GraphConstructor.py:9


def position_encoder(d_model: int, max_len: int = 20) -> torch.Tensor:
    table = torch.zeros(max_len, d_model)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq)
    return table


def get_voxel_floor_level(vfc: torch.Tensor, vbi: torch.Tensor) -> torch.Tensor:
    zeros = torch.zeros(vfc.shape[0], dtype=torch.long, device=vfc.device)
    first_floor = -max_pool_x(cluster=vbi, x=-vfc, batch=zeros)[0]
    return vfc - first_floor.index_select(0, vbi)


def MLP(dims, act: str = "") -> nn.Sequential:
    layers = []
    for a, b in zip(dims[:-1], dims[1:]):
        layers.append(nn.Linear(a, b, bias=True))
        if "leaky" in act:
            layers.append(nn.LeakyReLU(negative_slope=0.01))
        elif "relu" in act:
            layers.append(nn.ReLU())
        elif "tanh" in act:
            layers.append(nn.Tanh())
        elif "sigmoid" in act:
            layers.append(nn.Sigmoid())
    return nn.Sequential(*layers)


def extract_pos(feature: torch.Tensor):
    return feature[:, 3:6], torch.cat((feature[:, 0:3], feature[:, 6:]), dim=-1)


class ProgramBlock(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.messageMLP = MLP([2 * hidden_dim, hidden_dim])
        self.updateMLP = MLP([3 * hidden_dim, hidden_dim], act="leaky")

    def forward(self, feature, edge_index, class_index, class_feature, batch_index):
        aggr = propagate_mean(lambda feature_i, feature_j: self.messageMLP(torch.cat((feature_i, feature_j), -1)),
                              edge_index, feature.size(0), feature=feature)
        pooled = avg_pool_x(class_index, feature, batch_index, size=NUMBER_OF_CLASS)[0] * class_feature.view(-1, 1)
        on_node = pooled.index_select(0, class_index)
        return self.updateMLP(torch.cat((feature, aggr, on_node), -1))


class ProgramGNN(nn.Module):
    def __init__(self, input_dim, hidden_dim, noise_dim, layer_num):
        super().__init__()
        self.layer_num = layer_num
        self.enc = MLP([input_dim + noise_dim, hidden_dim])
        self.block = ProgramBlock(hidden_dim)

    def forward(self, cf, slf, e, cc, tr, cfb, z):
        x = self.enc(torch.cat([cf, slf.view(-1, 1), z], -1))
        for _ in range(self.layer_num):
            x = x + self.block(x, e, cc, tr, cfb)
        return x


class Attention(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.dec = MLP([hidden_dim, hidden_dim // 2, 2])
        self.NN_v = nn.Linear(hidden_dim, hidden_dim)
        self.NN_p = nn.Linear(hidden_dim, hidden_dim)
        self.theta = nn.Parameter(torch.Tensor(1, hidden_dim))
        glorot_(self.theta)

    def forward(self, program_graph_feature, voxel_feature, cep, cev, noise=None):
        exp_mask, exp_att = (None, None) if noise is None else noise
        soft_mask = dense_gumbel_softmax(self.dec(voxel_feature), exp_mask)
        hard_mask = straight_through_one_hot(soft_mask, -1)
        out_mask = {"hard": hard_mask[:, 0].view(-1, 1), "soft": soft_mask[:, 0].view(-1, 1)}

        p_j = program_graph_feature.index_select(0, cep)
        v_i = voxel_feature.index_select(0, cev)
        att = (self.theta.view(1, -1) * torch.tanh(self.NN_v(v_i) + self.NN_p(p_j))).sum(-1)
        soft_att, hard_att = segment_gumbel_softmax(att, cev, exp_noise=exp_att)
        out_att = {"hard": hard_att.view(-1, 1), "soft": soft_att.view(-1, 1)}

        weighted = out_att["soft"] * program_graph_feature[cep]
        new_voxel = voxel_feature + out_mask["soft"] * scatter_sum(weighted, cev, voxel_feature.shape[0])
        return out_mask, out_att, new_voxel

    @staticmethod
    def construct_output(program_class_feature, num_voxel, att, mask, cep, cev):
        idx = find_max_out_program_index(att["hard"] * mask["hard"].index_select(0, cev), cev, cep, num_voxel)
        hard_label = mask["hard"] * scatter_sum(att["hard"] * program_class_feature[cep], cev, num_voxel)
        soft_label = mask["hard"] * scatter_sum(att["soft"] * program_class_feature[cep], cev, num_voxel)
        return hard_label, soft_label, idx.view(-1)


class VoxelBlock(nn.Module):
    def __init__(self, hidden_dim, if_agg_story_in_update=True):
        super().__init__()
        self.if_agg_story_in_update = if_agg_story_in_update
        self.messageMLP = MLP([2 * hidden_dim + 3, hidden_dim])
        self.updateMLP = (MLP([4 * hidden_dim, hidden_dim], act="leaky") if if_agg_story_in_update
                          else MLP([2 * hidden_dim, hidden_dim]))

    def forward(self, voxel_feature, edge_index, batch_index, position, floor_index):
        def message(voxel_feature_i, voxel_feature_j, position_i, position_j):
            return self.messageMLP(torch.cat((voxel_feature_i, voxel_feature_j, position_i - position_j), -1))

        aggr = propagate_mean(message, edge_index, voxel_feature.size(0),
                              voxel_feature=voxel_feature, position=position)
        if not self.if_agg_story_in_update:
            return self.updateMLP(torch.cat((voxel_feature, aggr), -1))
        story, story_batch = avg_pool_x(floor_index, voxel_feature, batch_index)
        zeros = torch.zeros(story.shape[0], dtype=torch.long, device=story.device)
        building = avg_pool_x(story_batch, story, zeros)[0]
        story = story.index_select(0, floor_index)
        building = building.index_select(0, batch_index)
        return self.updateMLP(torch.cat((voxel_feature, aggr, story, building - story), -1))


class VoxelGNN_G(nn.Module):
    def __init__(self, input_dim, hidden_dim, noise_dim, layer_num, max_len: int = 20):
        super().__init__()
        self.layer_num = layer_num
        self.noise_dim = noise_dim
        self.register_buffer("pe", position_encoder(hidden_dim, max_len))
        self.enc = MLP([input_dim - 3 + noise_dim, hidden_dim])
        self.block = VoxelBlock(hidden_dim, if_agg_story_in_update=True)
        self.attention = Attention(hidden_dim)

    def forward(self, x, v, z, e, cep, cev, vbi, vfc, vfb, noise: Optional[Sequence] = None):
        draws = iter(noise) if noise is not None else None
        pos, v = extract_pos(v)
        v = self.enc(torch.cat((v, z), -1)) if self.noise_dim != 0 else self.enc(v)
        v = v + self.pe[get_voxel_floor_level(vfc, vbi), :]
        _, _, v = self.attention(x, v, cep, cev, None if draws is None else next(draws))
        out_mask = out_att = None
        for i in range(self.layer_num):
            v = v + self.block(v, e, vfb, pos, vfc)
            if i % 2 == 0 and i != 0:
                out_mask, out_att, v = self.attention(x, v, cep, cev, None if draws is None else next(draws))
        return out_mask, out_att, v


class VoxelGNN_D(nn.Module):
    def __init__(self, input_feature_dim, input_label_dim, hidden_dim, layer_num, max_len: int = 20):
        super().__init__()
        self.layer_num = layer_num
        self.register_buffer("pe", position_encoder(hidden_dim, max_len))
        self.feature_enc = MLP([input_feature_dim - 3, hidden_dim])
        self.label_enc = MLP([input_label_dim, hidden_dim])
        self.block = VoxelBlock(2 * hidden_dim, if_agg_story_in_update=False)

    def forward(self, v, l, e, vbi, vfc, vfb):
        pos, v = extract_pos(v)
        v = self.feature_enc(v)
        v = v + self.pe[get_voxel_floor_level(vfc, vbi), :]
        v = torch.cat([v, self.label_enc(l)], -1)
        for _ in range(self.layer_num):
            v = v + self.block(v, e, vfb, pos, vfc)
        return v


class Generator(nn.Module):
    def __init__(self, program_input_dim, voxel_input_dim, hidden_dim, noise_dim, program_layer, voxel_layer,
                 device=None, max_len: int = 20):
        super().__init__()
        self.device = device
        self.hidden_dim = hidden_dim
        self.noise_dim = noise_dim
        self.programGNN = ProgramGNN(program_input_dim, hidden_dim, noise_dim, program_layer)
        self.voxelGNN = VoxelGNN_G(voxel_input_dim, hidden_dim, 0, voxel_layer, max_len)

    def forward(self, graph, program_z, voxel_z, noise: Optional[Sequence] = None):
        x = self.programGNN(graph.program_class_feature, graph.story_level_feature, graph.program_edge,
                            graph.program_class_cluster, graph.program_target_ratio,
                            graph.program_class_feature_batch, program_z)
        cep, cev = graph.cross_edge_program_index_select, graph.cross_edge_voxel_index_select
        mask, att, _ = self.voxelGNN(x, graph.voxel_feature, voxel_z, graph.voxel_edge, cep, cev,
                                     graph.voxel_feature_batch, graph.voxel_floor_cluster,
                                     graph.voxel_feature_batch, noise)
        out, soft_out, idx = Attention.construct_output(graph.program_class_feature, graph.voxel_feature.shape[0],
                                                        att, mask, cep, cev)
        return out, soft_out, mask, att, idx


class DiscriminatorVoxel(nn.Module):
    def __init__(self, voxel_input_feature_dim, voxel_input_label_dim, hidden_dim, voxel_layer, act="",
                 max_len: int = 20):
        super().__init__()
        self.voxelGNN_D = VoxelGNN_D(voxel_input_feature_dim, voxel_input_label_dim, hidden_dim, voxel_layer, max_len)
        self.building_dec = MLP([2 * hidden_dim, hidden_dim, 1], act)
        self.story_dec = MLP([2 * hidden_dim, hidden_dim, 1], act)

    def forward(self, graph, out_label):
        v = self.voxelGNN_D(graph.voxel_feature, out_label, graph.voxel_edge, graph.voxel_feature_batch,
                            graph.voxel_floor_cluster, graph.voxel_feature_batch)
        story = global_max_pool(v, graph.voxel_floor_cluster)
        building = global_max_pool(v, graph.voxel_feature_batch)
        return self.building_dec(building), self.story_dec(story)
