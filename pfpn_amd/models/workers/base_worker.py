"""Optimiser construction for the actor-critic workers.

Host-side mirror of reference models/workers/base_worker.py: `ActorCriticWorker.build_optimizer`
(:25-76: one Adam when lr_actor == lr_critic and not separate_optimizer, else a critic Adam then an
actor Adam), `clip_grads` (:78-120: one joint tf.clip_by_global_norm), `train` logging (:136-150).
The resulting "optimizer op" is a SyncReplicasApplier (models/sync_model.py).
"""
import time

import numpy as np

from ..sync_model import SyncReplicasApplier, SyncWorker


def actor_critic_worker_wrapper(base_distributed_worker_class):
    class ActorCriticWorker(base_distributed_worker_class):
        def __init__(self, lr_critic, lr_actor, separate_optimizer=False, norm_clip=None, **kwargs):
            super().__init__(**kwargs)
            self.lr_critic = lr_critic
            self.lr_actor = lr_actor
            self.separate_optimizer = separate_optimizer
            if hasattr(norm_clip, "__len__"):
                raise NotImplementedError("per-network norm_clip dicts are unused by the PFPN settings")
            self.norm_clip = norm_clip
            self.log_interval_time = 10.0
            self.last_notify_time = time.perf_counter() - self.log_interval_time

        def build_optimizer(self, net):
            lr_a = self.lr_actor() if callable(self.lr_actor) else self.lr_actor
            lr_c = self.lr_critic() if callable(self.lr_critic) else self.lr_critic
            if self.separate_optimizer or lr_c != lr_a:
                # critic variables first, then actor variables (base_worker.py:33-45, :52-66)
                groups = [{"names": net.critic_param_names(), "lr": lr_c},
                          {"names": net.actor_param_names(), "lr": lr_a}]
            else:
                groups = [{"names": None, "lr": lr_c}]
            return SyncReplicasApplier(net, groups, norm_clip=self.norm_clip, world_size=self.n_workers
                                       if getattr(self, "n_workers", 1) > 1 else 1)

        def train(self, *args, **kwargs):
            loss, result = super().train(*args, **kwargs)
            if loss is not None:
                self.report([(loss, result)])
            return loss, result

        def report(self, losses):
            """[TRAIN] line of the chief: at most one per `log_interval_time`, always when a loss went NaN
            (reference base_worker.py:136-150); `losses` = the (loss tuple, fetches) pairs of the steps just taken."""
            if not (self.is_chief and losses):
                return
            now = time.perf_counter()
            nan = next((l for l, _ in losses if np.isnan(l[2]) or np.isnan(l[3])), None)
            if nan is None and now <= self.log_interval_time + self.last_notify_time:
                return
            _, policy_entropy, policy_loss, value_loss = nan if nan is not None else losses[-1][0]
            if self.log_dir:              # TensorBoard scalars under the reference's tags (actor_critic.py:132-184)
                if self.logger is None:
                    from ...summary import Summary
                    self.logger = Summary(self.log_dir)
                self.logger.add_scalar("loss/value_loss", value_loss, self.global_step)
                self.logger.add_scalar("loss/policy_loss", policy_loss, self.global_step)
                if policy_entropy is not None:
                    self.logger.add_scalar("loss/policy_entropy", policy_entropy, self.global_step)
                self.logger.flush()
            print("[TRAIN] Step: {}; Value Loss: {:.4f}; Policy Loss: {:.4f}; Policy Entropy: {:.4f}; "
                  "Worker: {}; {}".format(self.global_step, value_loss, policy_loss,
                                          0.0 if policy_entropy is None else policy_entropy, self.name,
                                          time.strftime("%m-%d %H:%M:%S", time.localtime())))
            self.last_notify_time = now

        @staticmethod
        def overtime(info):
            return "TimeLimit.truncated" in info and info["TimeLimit.truncated"]

    return ActorCriticWorker


SyncActorCriticWorker = actor_critic_worker_wrapper(SyncWorker)
