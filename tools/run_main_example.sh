set -e
R="2,2,2,2,2,2,"; E="96,96,96,96,96,96,"; W="2,4,8,2,4,8,2,4,8,2,4,8,2,4,8,2,4,8,"; D="1,1,1,1,1,1,"; NH="6,6,6,6,6,6,"; ML="4,4,4,4,4,4,"; Z="0,0,0,0,0,0,"; P="0.1,0.1,0.1,0.1,0.1,0.1,"   # README training flags: all three drop rates 0.1
COMMON="--arch tatt --mask --gradient --stu_iter_b1 3 --stu_iter_b2 3 --patch_size $R --embed_dim $E --window_size $W --depths $D --num_heads $NH --mlp_ratio $ML  --batch_size 8 --synthetic_steps 20"
timeout 300 python main.py $COMMON --drop_rate $P --attn_drop_rate $P --drop_path_rate $P --rotate_train 5 2>&1 | tail -3
timeout 300 python main.py $COMMON --drop_rate $P --attn_drop_rate $P --drop_path_rate $P --test 2>&1 | tail -2
cat ckpt/test_result.csv
