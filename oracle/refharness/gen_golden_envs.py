"""Golden fixtures for the other registered environments, from the UNMODIFIED reference:
CyberBattleTiny, ActiveDirectoryTiny, ActiveDirectory seeds 0 and 2, and the random traffic network for fixed seeds.

Run in the build container only (needs /root/reference):   python oracle/refharness/gen_golden_envs.py
Output, data only: tests/golden/topology_{tiny,tinyad,ad0,ad2,random_s1,random_s4,random_s5,random_s9,labelled_s4}.{bin,json} (blobs flattened from the
reference's own objects: they pin marlon_amd/samples/{tinytoy,active_directory,generate_network}.py) and step traces in the
format of gen_golden.py.  The reference's `new_environment` of the random network takes no seed (np.random.seed(None));
the fixtures call its two stages with an explicit seed: generate_random_traffic_network(seed=S, <new_environment's
arguments>), random.seed(S), cyberbattle_model_from_traffic_graph(<new_environment's arguments>).
"""
from __future__ import annotations

import json
import os
import random

import numpy as np

import gen_golden as G     # imports the reference through ref_loader
from marlon_amd import flatten as F

ref = G.ref


def ref_random_environment(seed: int, n_servers: int = 15):
    from cyberbattle.simulation import generate_network as gn
    traffic = gn.generate_random_traffic_network(
        seed=seed, n_clients=50, n_servers={"SMB": n_servers, "HTTP": n_servers, "RDP": n_servers},
        alpha=np.array([(1, 1), (0.2, 0.5)], dtype=float), beta=np.array([(1000, 10), (10, 100)], dtype=float))
    random.seed(seed)
    net = gn.cyberbattle_model_from_traffic_graph(
        traffic, cached_rdp_password_probability=0.8, cached_smb_password_probability=0.7,
        cached_accessed_network_shares_probability=0.8, cached_password_has_changed_probability=0.01,
        probability_two_nodes_use_same_password_to_access_given_resource=0.9)
    return ref.model.Environment(network=net, vulnerability_library=dict([]), identifiers=gn.ENV_IDENTIFIERS)


def main():
    from cyberbattle.samples.toyctf import tinytoy
    from cyberbattle.samples.active_directory import generate_ad, tiny_ad
    AG, DC, DG = ref.env.AttackerGoal, ref.env.DefenderConstraint, ref.env.DefenderGoal
    SAR = ref.defender.ScanAndReimageCompromisedMachines
    Env = ref.env.CyberBattleEnv

    builders = {
        "tiny": tinytoy.new_environment,
        "tinyad": tiny_ad.new_environment,
        "ad0": lambda: generate_ad.new_random_environment(0),
        "ad2": lambda: generate_ad.new_random_environment(2),
        "ad1": lambda: generate_ad.new_random_environment(1),      # 365 leakable credentials: the wide cached-credential set
        "ad6": lambda: generate_ad.new_random_environment(6),      # 821, the largest of ActiveDirectory-v0..v9
        "random_s1": lambda: ref_random_environment(1),     # its entry node has no outgoing traffic: no vulnerability at all (blob only)
        "random_s4": lambda: ref_random_environment(4),
        "random_s5": lambda: ref_random_environment(5),
        "random_s9": lambda: ref_random_environment(9),     # 38 nodes: the block model left many clients without any edge
    }
    import networkx as nx
    from marlon_amd.samples import labelled_graph
    builders["labelled_s4"] = lambda: labelled_graph.build(ref.model, 4, 6, graph=nx.path_graph(6, create_using=nx.DiGraph))
    topos = {k: F.flatten(b()) for k, b in builders.items()}
    for k, t in topos.items():
        with open(os.path.join(G.GOLDEN, f"topology_{k}.bin"), "wb") as f:
            f.write(t.blob)
        with open(os.path.join(G.GOLDEN, f"topology_{k}.json"), "w") as f:
            json.dump(dict(node_ids=t.node_ids, ports=t.ports, properties=t.properties,
                           local_vulnerabilities=t.local_vulnerabilities, remote_vulnerabilities=t.remote_vulnerabilities,
                           credential_strings=t.credential_strings, triples=t.triples), f, indent=0)
        print(f"topology_{k}: {t.n_nodes} nodes, {len(t.triples)} triples, {len(t.blob)} bytes")

    # ---- environments as the reference serialises them (model.py:545-554; yaml.dump sorts mappings, so the environment that
    # yaml.load returns lists its nodes alphabetically): the text, and the blob of what the reference reads back from it ----
    import yaml
    from marlon_amd.samples import kitchen_sink
    ref.model.setup_yaml_serializer()
    for k, env in (("toyctf", ref.toy_ctf.new_environment()), ("sink", kitchen_sink.build(ref.model))):
        text = yaml.dump(env)
        with open(os.path.join(G.GOLDEN, f"env_{k}.yaml"), "w") as f:
            f.write(text)
        with open(os.path.join(G.GOLDEN, f"topology_yaml_{k}.bin"), "wb") as f:
            f.write(F.flatten(yaml.load(text, yaml.Loader)).blob)

    def goal(**kw):
        g = dict(reward=0.0, low_availability=1.0, own_atleast=0, own_atleast_percent=1.0)
        g.update(kw)
        return g

    def spec(N, C, K, **kw):
        s = dict(maximum_node_count=N, maximum_total_credentials=C, maximum_discoverable_credentials_per_action=K,
                 attacker_goal=goal(), winning_reward=5000.0, losing_reward=0.0, maintain_sla=0.0, defender=None)
        s.update(kw)
        return s

    # ---- CyberBattleTiny-v0: registered kwargs (own_atleast=6 can never be met on 3 nodes: episodes end by eviction only),
    # plus an own-everything goal with a defender so that wins, re-imaging and eviction all occur ----
    sp = spec(10, 10, 5, attacker_goal=goal(own_atleast=6))
    G.run_trace("tiny_registered_s61", lambda: Env(tinytoy.new_environment(), attacker_goal=AG(own_atleast=6), defender_goal=DG(eviction=True),
                                                   maximum_total_credentials=10, maximum_node_count=10, throws_on_invalid_actions=False),
                topos["tiny"], 200, "mix", 61, sp)
    sp = spec(4, 3, 5, defender=["scan_and_reimage", 0.5, 1, 2], maintain_sla=0.3)
    G.run_trace("tiny_defender_s62", lambda: Env(tinytoy.new_environment(), attacker_goal=AG(own_atleast_percent=1.0), defender_agent=SAR(0.5, 1, 2),
                                                 defender_constraint=DC(maintain_sla=0.3), maximum_total_credentials=3, maximum_node_count=4,
                                                 throws_on_invalid_actions=False),
                topos["tiny"], 300, "valid", 62, sp, tape_dps=2)

    # ---- ActiveDirectoryTiny-v0 / ActiveDirectory-v0, -v2 (bounds that just fit instead of the registered 50 000) ----
    def ad_case(name, key, build, seed, steps, policy, K):
        t = topos[key]
        N, C = t.n_nodes, max(1, len(t.triples))
        G.run_trace(name, lambda: Env(build(), attacker_goal=AG(own_atleast_percent=1.0), maximum_total_credentials=C, maximum_node_count=N,
                                      maximum_discoverable_credentials_per_action=K, throws_on_invalid_actions=False),
                    t, steps, policy, seed, spec(N, C, K), store_masks=False)
    ad_case("tinyad_mix_s63", "tinyad", tiny_ad.new_environment, 63, 300, "mix", 20)
    ad_case("ad0_valid_s64", "ad0", lambda: generate_ad.new_random_environment(0), 64, 250, "valid", len(topos["ad0"].triples))
    ad_case("ad2_mix_s65", "ad2", lambda: generate_ad.new_random_environment(2), 65, 250, "mix", len(topos["ad2"].triples))
    ad_case("ad1_valid_s69", "ad1", lambda: generate_ad.new_random_environment(1), 69, 300, "valid", len(topos["ad1"].triples))
    ad_case("ad6_mix_s70", "ad6", lambda: generate_ad.new_random_environment(6), 70, 300, "mix", len(topos["ad6"].triples))

    # ---- model.assign_random_labels on a directed path, entry node at Admin privilege ----
    t = topos["labelled_s4"]
    G.run_trace("labelled_s4_mix_s68", lambda: Env(builders["labelled_s4"](), attacker_goal=AG(own_atleast_percent=1.0), maximum_total_credentials=3,
                                                   maximum_node_count=6, throws_on_invalid_actions=False),
                t, 200, "mix", 68, spec(6, 3, 5))

    # ---- ExternalRandomEvents (defender.py:58-148) on ToyCtf, Chain-4 and the kitchen sink (library vulnerabilities to plant):
    # vulnerabilities patched / planted, services stopped, firewall rules removed / added, every step, on every node ----
    ERE = ref.defender.ExternalRandomEvents
    from marlon_amd.samples import kitchen_sink as ks
    for name, key, make, N, C, seed, steps, sla in (
            ("toyctf_randomevents_s81", "toyctf", ref.toy_ctf.new_environment, 12, 10, 81, 400, 0.5),
            ("chain4_randomevents_s82", "chain4", lambda: ref.chainpattern.new_environment(4), 6, 6, 82, 400, 0.0),
            ("sink_randomevents_s83", "sink", lambda: ks.build(ref.model), None, None, 83, 400, 0.4)):
        t = F.flatten(make())
        N = N or t.n_nodes
        C = C or max(1, len(t.triples))
        G.run_trace(name, lambda make=make, N=N, C=C, sla=sla: Env(make(), attacker_goal=AG(own_atleast_percent=1.0), defender_agent=ERE(),
                                                                   defender_constraint=DC(maintain_sla=sla), maximum_total_credentials=C,
                                                                   maximum_node_count=N, throws_on_invalid_actions=False),
                    t, steps, "mix", seed, spec(N, C, 5, defender=["random_events"], maintain_sla=sla), tape_dps=12 * t.n_nodes + 8)

    # ---- random traffic network (CyberBattleRandom-v0's generator, seeded), attacker only and with a defender ----
    t = topos["random_s4"]
    N, C = t.n_nodes, len(t.triples)
    G.run_trace("random_s4_valid_s66", lambda: Env(ref_random_environment(4), attacker_goal=AG(own_atleast_percent=1.0), maximum_total_credentials=C,
                                                   maximum_node_count=N, maximum_discoverable_credentials_per_action=32, throws_on_invalid_actions=False),
                t, 400, "valid", 66, spec(N, C, 32), store_masks=False)
    t = topos["random_s5"]
    N, C = t.n_nodes, len(t.triples)
    G.run_trace("random_s5_defender_s67", lambda: Env(ref_random_environment(5), attacker_goal=AG(own_atleast_percent=1.0), defender_agent=SAR(0.6, 4, 3),
                                                      defender_constraint=DC(maintain_sla=0.5), maximum_total_credentials=C, maximum_node_count=N,
                                                      maximum_discoverable_credentials_per_action=32, throws_on_invalid_actions=False),
                t, 300, "mix", 67, spec(N, C, 32, defender=["scan_and_reimage", 0.6, 4, 3], maintain_sla=0.5), tape_dps=8, store_masks=False)


if __name__ == "__main__":
    main()
