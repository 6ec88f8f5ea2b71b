set pagination off
set confirm off
set breakpoint pending on
set height 0
break cdkf_custom_kernel
run
delete 1
break *(&cdkf_custom_kernel + 0x2DA40)
commands
silent
printf "FINAL(good build) total(s1:s0)=%#x:%#x exec=%#lx gid(lane0)=%#x:%#x p(v28 lane0)=%d g(lane0)=%#x:%#x\n", $s1, $s0, $exec, $v3[0], $v2[0], $v28[0], $v7[0], $v6[0]
continue
end
continue
quit
