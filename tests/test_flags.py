from distributed_tensorflow_amd import flags


def test_demo2_reference_defaults():
    F, rest = flags.parse_known(flags.build_mnist_parser(True), [])
    assert F.ps_hosts == "192.168.1.106:2222"
    assert F.worker_hosts == "192.168.1.104:2223, 192.168.1.105:2224"
    assert F.job_name == "ps" and F.task_index == 0
    assert F.batch_size == 100 and F.learning_rate == 1e-4 and F.training_steps == 10000
    assert F.keep_prob == 0.7 and F.eval_step_interval == 100 and F.save_model_secs == 600
    assert F.double_softmax is True and F.sync is True
    assert flags.parse_host_list(F.worker_hosts) == ["192.168.1.104:2223", "192.168.1.105:2224"]  # Q10


def test_retrain_reference_defaults():
    F1, _ = flags.parse_known(flags.build_retrain_parser(False), [])
    F2, _ = flags.parse_known(flags.build_retrain_parser(True), [])
    assert F1.training_steps == 10000 and F2.training_steps == 2000
    assert F2.ps_hosts == "192.168.1.105:2222" and F2.job_name == "worker"
    assert F2.worker_hosts == "192.168.1.104:2223, 192.168.1.106:2224"
    for F in (F1, F2):
        assert F.image_dir == "./data" and F.output_graph == "./retrained_graph.pb"
        assert F.output_labels == "./retrained_labels.txt" and F.summaries_dir == "./retrain_logs"
        assert F.learning_rate == 0.01 and F.testing_percentage == 10 and F.validation_percentage == 10
        assert F.eval_step_interval == 10 and F.train_batch_size == 100 and F.test_batch_size == -1
        assert F.validation_batch_size == 100 and F.print_misclassified_test_images is False
        assert F.model_dir == "./inception_model" and F.bottleneck_dir == "./bottlenecks"
        assert F.final_tensor_name == "final_result" and F.flip_left_right is False
        assert F.random_crop == 0 and F.random_scale == 0 and F.random_brightness == 0


def test_unknown_flags_are_tolerated_like_parse_known_args():
    F, rest = flags.parse_known(flags.build_mnist_parser(True), ["--job_name=worker", "--task_index", "3", "--bogus", "1"])
    assert F.job_name == "worker" and F.task_index == 3 and rest == ["--bogus", "1"]
    F, _ = flags.parse_known(flags.build_mnist_parser(True), ["--async", "--double_softmax=false"])
    assert F.sync is False and F.double_softmax is False
